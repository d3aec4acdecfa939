// malcs_diag.hip -- kernel family 1, window class: distances of similar pairs from a few limbs around
// the main diagonal.
//
// Stands in for edit_distance(a, b) (reference algebra/lcs/edit_distance.py:21-53).  The reference's
// O(NP) sweep only visits the diagonals an alignment of cost <= D can reach; this is the bit-vector
// counterpart: an alignment of cost <= T stays on diagonals k = col - row in
// [-(T - delta)/2, delta + (T - delta)/2] (delta = n - m >= 0), so of every column only
// T + 1 rows matter.  Rows are handled in limbs of 32; a window of NLW limbs that moves down one limb
// per block of 32 columns covers all of them when T = 32 * NLW - 32.  The window is anchored on the columns:
// block i is columns [32 i, 32 i + 32) of the text and rows [32 i - kmax, 32 i - kmax + 32 NLW) of the pattern
// (kmax = the band's top diagonal, per pair), so every block of a pair but its last has all 32 columns.
//
//   * one LANE per pair, 64 pairs per wave, no cross-lane traffic at all: the window's limbs live in
//     VGPRs (state V, pattern bit planes P), the carry chain of a column runs through the lane's own
//     limbs (v_addc_co_u32 on SGPR-pair lane masks, as in malcs_dist.hip);
//   * after 32 columns the lowest limb leaves the window (its zero bits are banked: they are that
//     limb's share of the LLCS), the others move down one register, a fresh limb (V = all ones) enters
//     on top.  Rows below the window keep what they had, rows above have not started: both restrict
//     the DP to paths inside the window, so the result D' >= D, and D' == D whenever D' <= T
//     (then no optimal alignment leaves the band);
//   * pairs with D' > T are appended to a device-side list per full-matrix launch class and redone
//     by k_llcs_bitpar_list (malcs_dist.hip) in the same launch sequence, no host round trip;
//   * data layout: at plan time the common ends of every pair are cut off (they do not change the distance),
//     the pairs of the class are sorted by what is left and cut into tiles of 64, and the tiles are dealt to
//     SIMDs by length (malcs_diag_plan); k_diag_layout writes, per tile, rows of 64 dwords (one per lane):
//     4 meta rows, the NLW-1 limbs the window starts with, then per block of columns the entering pattern
//     limb, already shifted to the lane's own diagonal offset, and the 32 text bits of that block.  The main
//     loop therefore reads 2 * NP fully coalesced rows (1 KB for DNA) per 32 columns and wave and does no
//     address arithmetic or funnel shifts of its own.
//
// Work per column and lane: NLW * (NP + 2) + NP VALU ops; for DNA and NLW = 3 that is 14 ops for what
// the full-matrix kernel spends 4 * ceil(m / 32) + 2 on (130 for 1 kb).
#include "malcs_internal.hpp"
#include "malcs_device.hpp"
#include "malcs_bitpar.hpp"
#include "malcs_hostpack.hpp"
#include <hip/hip_ext.h>
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <system_error>
#include <thread>
#include <utility>

namespace {

using namespace malcs_dev;

constexpr int META_ROWS = 4;          // pid, m, n, kmax | retry class << 16
constexpr uint32_t JOIN_WORDS = 3;        // 64-bit words per lane a cut tile has in the join buffer: the word the parts swap, the front part's rows, the back part's
constexpr int MALCS_DIAG_HALF_T = 48; // two limbs moving 16 rows per 16 columns: 64 - 16 diagonals

#include "malcs_plan.inc"

// ---- one block of 32 columns, skewed inside the lane ---------------------------------------------
// A wave issues in order, and the straightforward order (limb after limb, column after column) makes
// every instruction wait for the one before it.  So the limbs of a lane are skewed like the lanes of the
// full-matrix kernel: in step s limb k works on column s - k.  The carry out of limb k then belongs to
// the column limb k + 1 handles in the NEXT step, and within a step the NLW limbs are independent:
// their instructions are interleaved (all first bit-ops, all second ones, the additions, all V'), which
// puts NLW - 1 independent instructions between an instruction and its consumer.  The skew is filled
// and drained once per block (NLW - 1 extra steps of 32 + NLW - 1), because the window moves between blocks.
// Every instruction is its own volatile asm statement: the order written here is the order issued.
__device__ __forceinline__ uint32_t op_match(uint32_t v, uint32_t p, uint32_t x)      // v & ~(p ^ x)
{
    uint32_t d;
    asm volatile("v_bitop3_b32 %0, %1, %2, %3 bitop3:0x90" : "=v"(d) : "v"(v), "v"(p), "v"(x));
    return d;
}
__device__ __forceinline__ uint32_t op_and(uint32_t a, uint32_t b)
{
    uint32_t d;
    asm volatile("v_and_b32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ uint32_t op_next(uint32_t u, uint32_t sum, uint32_t v)     // sum | (v ^ u)
{
    uint32_t d;
    asm volatile("v_bitop3_b32 %0, %1, %2, %3 bitop3:0xde" : "=v"(d) : "v"(u), "v"(sum), "v"(v));
    return d;
}
__device__ __forceinline__ uint32_t op_add(uint32_t a, uint32_t b, uint64_t &cout)
{
    uint32_t d;
    asm volatile("v_add_co_u32_e64 %0, %1, %2, %3" : "=v"(d), "=s"(cout) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ uint32_t op_addc(uint32_t a, uint32_t b, uint64_t cin, uint64_t &cout)
{
    uint32_t d;
    asm volatile("v_addc_co_u32_e64 %0, %1, %2, %3, %4" : "=v"(d), "=&s"(cout) : "v"(a), "v"(b), "s"(cin));
    return d;
}

template <int NLW, int NP>
struct diag_state {
    uint32_t x[NLW][NP];      // text bits (sign-extended) of the column each limb works on in this step
    uint32_t act[NLW];        // ... and whether that column exists (masked blocks only)
    uint64_t c[NLW + 1];      // c[k]: carry into limb k for the column it handles next (c[0] is never read)
};

template <int NLW, int NP, bool MASKED, int S>
__device__ __forceinline__ void diag_step(uint32_t (&V)[NLW], const uint32_t (&P)[NLW][NP], const uint32_t (&w)[NP],
                                          uint32_t vm, diag_state<NLW, NP> &st)
{
    constexpr int klo = S < 32 ? 0 : S - 31;                  // limbs with a column in this step
    constexpr int khi = S < NLW - 1 ? S : NLW - 1;
#pragma unroll
    for (int k = NLW - 1; k >= 1; k--) {
#pragma unroll
        for (int q = 0; q < NP; q++) st.x[k][q] = st.x[k - 1][q];
        st.act[k] = st.act[k - 1];
    }
    if constexpr (S < 32) {
#pragma unroll
        for (int q = 0; q < NP; q++) st.x[0][q] = (uint32_t)__builtin_amdgcn_sbfe((int)w[q], S, 1);
        if (MASKED) st.act[0] = (uint32_t)__builtin_amdgcn_sbfe((int)vm, S, 1);
    }
    uint32_t t[NLW], sum[NLW];
#pragma unroll
    for (int q = 0; q < NP; q++) {
#pragma unroll
        for (int k = klo; k <= khi; k++) t[k] = op_match(q ? t[k] : V[k], P[k][q], st.x[k][q]);
    }
    if (MASKED) {
#pragma unroll
        for (int k = klo; k <= khi; k++) t[k] = op_and(t[k], st.act[k]);
    }
#pragma unroll
    for (int k = khi; k >= klo; k--) {                        // descending: c[k + 1] is consumed before it is rewritten
        if (k == 0) sum[0] = op_add(V[0], t[0], st.c[1]);
        else        sum[k] = op_addc(V[k], t[k], st.c[k], st.c[k + 1]);
    }
#pragma unroll
    for (int k = klo; k <= khi; k++) V[k] = op_next(t[k], sum[k], V[k]);
}

// DNA, three limbs (the default class): the whole unmasked block as ONE asm statement, so that neither
// compiler pads nor its scheduling get between the instructions.  Temporaries are named registers
// (clobbered) and the operands are pinned, all chosen so that the three sources of every instruction sit in
// three different VGPR banks (bank = register number mod 4; tools/diag_micro.hip: 55 T lane-op/s against
// 42 T when they collide): V v20/v24/v28 and the text words v32/v36 in bank 0, the pattern planes
// v21/v25/v29 and v33/v37/v41 in bank 1, U per limb v22/v26/v30 in bank 2, the sums v23/v27/v31 and the
// three rotating pairs of text bits v35,v47 / v39,v51 / v43,v55 in bank 3;
// the carries into limbs 1 / 2 alternate between s[40:41] / s[42:43] (written by even steps) and
// s[44:45] / s[46:47] (odd steps), s[48:49] swallows the carry out of the window.
// Step s: limb 0 on column s (text bits in pair A), limb 1 on column s - 1 (B), limb 2 on column s - 2 (C).
#define DG_X0 "v35", "v47"
#define DG_X1 "v39", "v51"
#define DG_X2 "v43", "v55"
#define DG_CE "s[40:41]", "s[42:43]"
#define DG_CO "s[44:45]", "s[46:47]"
#define DG_BFE(s, A0, A1) "v_bfe_i32 " A0 ", %[w0], " #s ", 1\n\tv_bfe_i32 " A1 ", %[w1], " #s ", 1\n\t"
#define DG_U0a(A0) "v_bitop3_b32 v22, %[V0], %[P00], " A0 " bitop3:0x90\n\t"
#define DG_U1a(B0) "v_bitop3_b32 v26, %[V1], %[P10], " B0 " bitop3:0x90\n\t"
#define DG_U2a(C0) "v_bitop3_b32 v30, %[V2], %[P20], " C0 " bitop3:0x90\n\t"
#define DG_U0b(A1) "v_bitop3_b32 v22, v22, %[P01], " A1 " bitop3:0x90\n\t"
#define DG_U1b(B1) "v_bitop3_b32 v26, v26, %[P11], " B1 " bitop3:0x90\n\t"
#define DG_U2b(C1) "v_bitop3_b32 v30, v30, %[P21], " C1 " bitop3:0x90\n\t"
#define DG_ADD2(I2) "v_addc_co_u32_e64 v31, s[48:49], %[V2], v30, " I2 "\n\t"
#define DG_ADD1(I1, O2) "v_addc_co_u32_e64 v27, " O2 ", %[V1], v26, " I1 "\n\t"
#define DG_ADD0(O1) "v_add_co_u32_e64 v23, " O1 ", %[V0], v22\n\t"
#define DG_N2 "v_bitop3_b32 %[V2], v30, v31, %[V2] bitop3:0xde\n\t"
#define DG_N1 "v_bitop3_b32 %[V1], v26, v27, %[V1] bitop3:0xde\n\t"
#define DG_N0 "v_bitop3_b32 %[V0], v22, v23, %[V0] bitop3:0xde\n\t"
#define DG_STEP_(s, A0, A1, B0, B1, C0, C1, I1, I2, O1, O2)                                              \
    DG_BFE(s, A0, A1) DG_U1a(B0) DG_U2a(C0) DG_U0a(A0) DG_U1b(B1) DG_U2b(C1) DG_U0b(A1)                  \
    DG_ADD2(I2) DG_ADD1(I1, O2) DG_ADD0(O1) DG_N2 DG_N1 DG_N0
#define DG_STEP_X(s, A0, A1, B0, B1, C0, C1, I1, I2, O1, O2) DG_STEP_(s, A0, A1, B0, B1, C0, C1, I1, I2, O1, O2)
#define DG_STEP(s, A, B, C, I, O) DG_STEP_X(s, A, B, C, I, O)
#define DG_FILL0_(s, A0, A1, O1, O2) DG_BFE(s, A0, A1) DG_U0a(A0) DG_U0b(A1) DG_ADD0(O1) DG_N0
#define DG_FILL0_X(s, A0, A1, O1, O2) DG_FILL0_(s, A0, A1, O1, O2)
#define DG_FILL0(s, A, O) DG_FILL0_X(s, A, O)
#define DG_FILL1_(s, A0, A1, B0, B1, I1, I2, O1, O2)                                                     \
    DG_BFE(s, A0, A1) DG_U1a(B0) DG_U0a(A0) DG_U1b(B1) DG_U0b(A1) DG_ADD1(I1, O2) DG_ADD0(O1) DG_N1 DG_N0
#define DG_FILL1_X(s, A0, A1, B0, B1, I1, I2, O1, O2) DG_FILL1_(s, A0, A1, B0, B1, I1, I2, O1, O2)
#define DG_FILL1(s, A, B, I, O) DG_FILL1_X(s, A, B, I, O)
#define DG_DRAIN1_(B0, B1, C0, C1, I1, I2, O1, O2) DG_U1a(B0) DG_U2a(C0) DG_U1b(B1) DG_U2b(C1) DG_ADD2(I2) DG_ADD1(I1, O2) DG_N2 DG_N1
#define DG_DRAIN1_X(B0, B1, C0, C1, I1, I2, O1, O2) DG_DRAIN1_(B0, B1, C0, C1, I1, I2, O1, O2)
#define DG_DRAIN1(B, C, I, O) DG_DRAIN1_X(B, C, I, O)
#define DG_DRAIN2_(C0, C1, I1, I2) DG_U2a(C0) DG_U2b(C1) DG_ADD2(I2) DG_N2
#define DG_DRAIN2_X(C0, C1, I1, I2) DG_DRAIN2_(C0, C1, I1, I2)
#define DG_DRAIN2(C, I) DG_DRAIN2_X(C, I)
// six steps from an even s with s % 3 == 2: the pairs rotate A = X[s % 3], B = X[(s - 1) % 3], C = X[(s - 2) % 3],
// even steps read the carries the odd ones wrote and the other way round
#define DG_SIX(s0, s1, s2, s3, s4, s5)                                                                    \
    DG_STEP(s0, DG_X2, DG_X1, DG_X0, DG_CO, DG_CE) DG_STEP(s1, DG_X0, DG_X2, DG_X1, DG_CE, DG_CO)        \
    DG_STEP(s2, DG_X1, DG_X0, DG_X2, DG_CO, DG_CE) DG_STEP(s3, DG_X2, DG_X1, DG_X0, DG_CE, DG_CO)        \
    DG_STEP(s4, DG_X0, DG_X2, DG_X1, DG_CO, DG_CE) DG_STEP(s5, DG_X1, DG_X0, DG_X2, DG_CE, DG_CO)

// The rows of the next block are loaded by the same statement (issued first, waited for last), because
// a compiler-visible load in flight makes hipcc drain the memory counter in front of the statement.
__device__ __forceinline__ void diag_block_dna3(uint32_t (&V)[3], const uint32_t (&P)[3][2], const uint32_t (&w)[2],
                                                const uint32_t *next_rows, uint32_t (&nxt)[4])
{
    register uint32_t V0 asm("v20") = V[0], V1 asm("v24") = V[1], V2 asm("v28") = V[2];
    register uint32_t P00 asm("v21") = P[0][0], P10 asm("v25") = P[1][0], P20 asm("v29") = P[2][0];
    register uint32_t P01 asm("v33") = P[0][1], P11 asm("v37") = P[1][1], P21 asm("v41") = P[2][1];
    register uint32_t w0 asm("v32") = w[0], w1 asm("v36") = w[1];
    asm volatile(
        "global_load_dword %[n0], %[np], off\n\tglobal_load_dword %[n1], %[np], off offset:256\n\t"
        "global_load_dword %[n2], %[np], off offset:512\n\tglobal_load_dword %[n3], %[np], off offset:768\n\t"
        ".p2align 3\n\ts_nop 0\n\t"             // 64-bit encodings issue a third faster from addresses = 4 (mod 8): tools/diag_micro.hip
        DG_FILL0(0, DG_X0, DG_CE) DG_FILL1(1, DG_X1, DG_X0, DG_CE, DG_CO)
        DG_SIX(2, 3, 4, 5, 6, 7) DG_SIX(8, 9, 10, 11, 12, 13) DG_SIX(14, 15, 16, 17, 18, 19)
        DG_SIX(20, 21, 22, 23, 24, 25) DG_SIX(26, 27, 28, 29, 30, 31)
        DG_DRAIN1(DG_X1, DG_X0, DG_CO, DG_CE) DG_DRAIN2(DG_X1, DG_CE)
        "s_waitcnt vmcnt(0)"
        : [V0] "+v"(V0), [V1] "+v"(V1), [V2] "+v"(V2),
          [n0] "=&v"(nxt[0]), [n1] "=&v"(nxt[1]), [n2] "=&v"(nxt[2]), [n3] "=&v"(nxt[3])
        : [np] "v"(next_rows), [P00] "v"(P00), [P01] "v"(P01), [P10] "v"(P10), [P11] "v"(P11), [P20] "v"(P20), [P21] "v"(P21),
          [w0] "v"(w0), [w1] "v"(w1)
        : "v22", "v26", "v30", "v23", "v27", "v31", "v35", "v47", "v39", "v51", "v43", "v55",
          "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49");
    V[0] = V0; V[1] = V1; V[2] = V2;
}

// Three blocks in one statement.  Between two blocks the window moves down one limb; done in registers
// that means a dozen moves (V, both pattern planes, the text words) next to mask arithmetic for the
// first / last blocks of a pair.  Inside the fast stretch of a tile (every lane's columns and leaving rows
// exist, see the kernel) none of that is needed: the three limbs trade ROLES instead of contents -- block A
// runs with limbs (a, b, c) as (0, 1, 2), block B with (b, c, a), block C with (c, a, b) -- so after three
// blocks every limb is back in its place.  The limb that leaves banks its zero bits (v_not + v_bcnt),
// restarts at all ones and takes the pattern planes of the limb that enters (loaded one block ahead into
// the four result registers, moved over after the wait).  12 instructions per transition instead of 56.
// Registers as above; limb a = V v20 / P v21, v33; b = v24 / v25, v37; c = v28 / v29, v41.
#define TG_LA "v20", "v21", "v33"
#define TG_LB "v24", "v25", "v37"
#define TG_LC "v28", "v29", "v41"
#define TG_Ua(U, V, P0, X0) "v_bitop3_b32 " U ", " V ", " P0 ", " X0 " bitop3:0x90\n\t"
#define TG_Ub(U, P1, X1) "v_bitop3_b32 " U ", " U ", " P1 ", " X1 " bitop3:0x90\n\t"
#define TG_AD0(S, O, V, U) "v_add_co_u32_e64 " S ", " O ", " V ", " U "\n\t"
#define TG_ADC(S, O, V, U, I) "v_addc_co_u32_e64 " S ", " O ", " V ", " U ", " I "\n\t"
#define TG_N(V, U, S) "v_bitop3_b32 " V ", " U ", " S ", " V " bitop3:0xde\n\t"
// roles: limb 0 works in v22 / v23, limb 1 in v26 / v27, limb 2 in v30 / v31 (U / sum)
#define TG_STEP_(s, A0, A1, B0, B1, C0, C1, I1, I2, O1, O2, V0, P00, P01, V1, P10, P11, V2, P20, P21)     \
    DG_BFE(s, A0, A1) TG_Ua("v26", V1, P10, B0) TG_Ua("v30", V2, P20, C0) TG_Ua("v22", V0, P00, A0)       \
    TG_Ub("v26", P11, B1) TG_Ub("v30", P21, C1) TG_Ub("v22", P01, A1)                                      \
    TG_ADC("v31", "s[48:49]", V2, "v30", I2) TG_ADC("v27", O2, V1, "v26", I1) TG_AD0("v23", O1, V0, "v22") \
    TG_N(V2, "v30", "v31") TG_N(V1, "v26", "v27") TG_N(V0, "v22", "v23")
#define TG_STEP(s, A, B, C, I, O, V0, P00, P01, V1, P10, P11, V2, P20, P21) \
    TG_STEP_(s, A, B, C, I, O, V0, P00, P01, V1, P10, P11, V2, P20, P21)
#define TG_FILL0_(s, A0, A1, O1, O2, V0, P00, P01) \
    DG_BFE(s, A0, A1) TG_Ua("v22", V0, P00, A0) TG_Ub("v22", P01, A1) TG_AD0("v23", O1, V0, "v22") TG_N(V0, "v22", "v23")
#define TG_FILL0(s, A, O, V0, P00, P01) TG_FILL0_(s, A, O, V0, P00, P01)
#define TG_FILL1_(s, A0, A1, B0, B1, I1, I2, O1, O2, V0, P00, P01, V1, P10, P11)                           \
    DG_BFE(s, A0, A1) TG_Ua("v26", V1, P10, B0) TG_Ua("v22", V0, P00, A0) TG_Ub("v26", P11, B1) TG_Ub("v22", P01, A1) \
    TG_ADC("v27", O2, V1, "v26", I1) TG_AD0("v23", O1, V0, "v22") TG_N(V1, "v26", "v27") TG_N(V0, "v22", "v23")
#define TG_FILL1(s, A, B, I, O, V0, P00, P01, V1, P10, P11) TG_FILL1_(s, A, B, I, O, V0, P00, P01, V1, P10, P11)
#define TG_DRAIN1_(B0, B1, C0, C1, I1, I2, O1, O2, V1, P10, P11, V2, P20, P21)                             \
    TG_Ua("v26", V1, P10, B0) TG_Ua("v30", V2, P20, C0) TG_Ub("v26", P11, B1) TG_Ub("v30", P21, C1)        \
    TG_ADC("v31", "s[48:49]", V2, "v30", I2) TG_ADC("v27", O2, V1, "v26", I1) TG_N(V2, "v30", "v31") TG_N(V1, "v26", "v27")
#define TG_DRAIN1(B, C, I, O, V1, P10, P11, V2, P20, P21) TG_DRAIN1_(B, C, I, O, V1, P10, P11, V2, P20, P21)
#define TG_DRAIN2_(C0, C1, I1, I2, V2, P20, P21) \
    TG_Ua("v30", V2, P20, C0) TG_Ub("v30", P21, C1) TG_ADC("v31", "s[48:49]", V2, "v30", I2) TG_N(V2, "v30", "v31")
#define TG_DRAIN2(C, I, V2, P20, P21) TG_DRAIN2_(C, I, V2, P20, P21)
#define TG_SIX(s0, s1, s2, s3, s4, s5, V0, P00, P01, V1, P10, P11, V2, P20, P21)                            \
    TG_STEP(s0, DG_X2, DG_X1, DG_X0, DG_CO, DG_CE, V0, P00, P01, V1, P10, P11, V2, P20, P21) TG_STEP(s1, DG_X0, DG_X2, DG_X1, DG_CE, DG_CO, V0, P00, P01, V1, P10, P11, V2, P20, P21) \
    TG_STEP(s2, DG_X1, DG_X0, DG_X2, DG_CO, DG_CE, V0, P00, P01, V1, P10, P11, V2, P20, P21) TG_STEP(s3, DG_X2, DG_X1, DG_X0, DG_CE, DG_CO, V0, P00, P01, V1, P10, P11, V2, P20, P21) \
    TG_STEP(s4, DG_X0, DG_X2, DG_X1, DG_CO, DG_CE, V0, P00, P01, V1, P10, P11, V2, P20, P21) TG_STEP(s5, DG_X1, DG_X0, DG_X2, DG_CE, DG_CO, V0, P00, P01, V1, P10, P11, V2, P20, P21)
#define TG_BLOCK_(V0, P00, P01, V1, P10, P11, V2, P20, P21)                                                 \
    ".p2align 3\n\ts_nop 0\n\t"                                                                             \
    TG_FILL0(0, DG_X0, DG_CE, V0, P00, P01) TG_FILL1(1, DG_X1, DG_X0, DG_CE, DG_CO, V0, P00, P01, V1, P10, P11) \
    TG_SIX(2, 3, 4, 5, 6, 7, V0, P00, P01, V1, P10, P11, V2, P20, P21) TG_SIX(8, 9, 10, 11, 12, 13, V0, P00, P01, V1, P10, P11, V2, P20, P21) TG_SIX(14, 15, 16, 17, 18, 19, V0, P00, P01, V1, P10, P11, V2, P20, P21) \
    TG_SIX(20, 21, 22, 23, 24, 25, V0, P00, P01, V1, P10, P11, V2, P20, P21) TG_SIX(26, 27, 28, 29, 30, 31, V0, P00, P01, V1, P10, P11, V2, P20, P21)                            \
    TG_DRAIN1(DG_X1, DG_X0, DG_CO, DG_CE, V1, P10, P11, V2, P20, P21) TG_DRAIN2(DG_X1, DG_CE, V2, P20, P21)
#define TG_BLOCK(L0, L1, L2) TG_BLOCK_(L0, L1, L2)
#define TG_LOADS(o0, o1, o2, o3)                                                                            \
    "global_load_dword %[n0], %[np], off offset:" #o0 "\n\tglobal_load_dword %[n1], %[np], off offset:" #o1 "\n\t" \
    "global_load_dword %[n2], %[np], off offset:" #o2 "\n\tglobal_load_dword %[n3], %[np], off offset:" #o3 "\n\t"
// the limb (V, P0, P1) leaves, restarts and takes the planes of the entering limb; the text words follow
#define TG_BANK_(V, P0, P1) "v_not_b32 v22, " V "\n\tv_bcnt_u32_b32 %[bank], v22, %[bank]\n\t"
#define TG_BANK_X(V, P0, P1) TG_BANK_(V, P0, P1)
#define TG_BANK(L) TG_BANK_X(L)
#define TG_ENTER_(V, P0, P1)                                                                                \
    "v_mov_b32 " V ", -1\n\ts_waitcnt vmcnt(0)\n\tv_mov_b32 " P0 ", %[n0]\n\tv_mov_b32 " P1 ", %[n1]\n\t"   \
    "v_mov_b32 v32, %[n2]\n\tv_mov_b32 v36, %[n3]\n\t"
#define TG_ENTER_X(V, P0, P1) TG_ENTER_(V, P0, P1)
#define TG_ENTER(L) TG_ENTER_X(L)

// Runs triples of blocks while `more(it)` holds.  On entry and on return nxt holds the rows of block `it`
// and p points at them; V[2] / P[2] are dead outside (the limb that enters is set up here, the one that
// has left is restarted by the caller).  The limbs stay in their pinned registers across the loop.
template <typename More>
__device__ __forceinline__ void diag_triples_dna3(uint32_t (&V)[3], uint32_t (&P)[3][2], uint32_t &banked,
                                                  const uint32_t *&p, uint32_t (&nxt)[4], int &it, More more)
{
    register uint32_t V0 asm("v20") = V[0], V1 asm("v24") = V[1], V2 asm("v28");
    register uint32_t P00 asm("v21") = P[0][0], P10 asm("v25") = P[1][0], P20 asm("v29");
    register uint32_t P01 asm("v33") = P[0][1], P11 asm("v37") = P[1][1], P21 asm("v41");
    register uint32_t w0 asm("v32"), w1 asm("v36");
    uint32_t n0 = nxt[0], n1 = nxt[1], n2 = nxt[2], n3 = nxt[3], bank = banked;
    do {
        asm volatile(
            "v_mov_b32 v29, %[n0]\n\tv_mov_b32 v41, %[n1]\n\tv_mov_b32 v32, %[n2]\n\tv_mov_b32 v36, %[n3]\n\tv_mov_b32 v28, -1\n\t"
            TG_LOADS(1024, 1280, 1536, 1792)
            TG_BLOCK(TG_LA, TG_LB, TG_LC)
            TG_BANK(TG_LA) TG_ENTER(TG_LA) TG_LOADS(2048, 2304, 2560, 2816)
            TG_BLOCK(TG_LB, TG_LC, TG_LA)
            TG_BANK(TG_LB) TG_ENTER(TG_LB) TG_LOADS(3072, 3328, 3584, 3840)
            TG_BLOCK(TG_LC, TG_LA, TG_LB)
            TG_BANK(TG_LC)
            "s_waitcnt vmcnt(0)"
            : "+v"(V0), "+v"(V1), "=&v"(V2), "+v"(P00), "+v"(P01), "+v"(P10), "+v"(P11), "=&v"(P20), "=&v"(P21),
              [w0] "=&v"(w0), [w1] "=&v"(w1), [bank] "+v"(bank), [n0] "+v"(n0), [n1] "+v"(n1), [n2] "+v"(n2), [n3] "+v"(n3)
            : [np] "v"(p)
            : "v22", "v26", "v30", "v23", "v27", "v31", "v35", "v47", "v39", "v51", "v43", "v55",
              "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49");
        p += 3 * 4 * 64;
        it += 3;
    } while (more(it));
    V[0] = V0; V[1] = V1;
    P[0][0] = P00; P[0][1] = P01; P[1][0] = P10; P[1][1] = P11;
    nxt[0] = n0; nxt[1] = n1; nxt[2] = n2; nxt[3] = n3;
    banked = bank;
}

// ---- two limbs, half-limb steps --------------------------------------------------------------------
// The distances of a similar batch need fewer diagonals than three limbs cover.  Two limbs that move down 16 rows per
// 16 columns cover T = 48 (64 rows - 16) for 10.75 instead of 14.5 instructions per column (tools/two_limb_rate.hip:
// 16.5 against 21.7 ns per column and wave at two waves per SIMD).  The rows are the ones the three-limb class reads
// (two limbs to start with, per block the limb that enters and the text word): the window of 64 rows lies in the
// pattern limbs a, b (first 16 columns of a block) and in a, b, c shifted by 16 (last 16; four v_alignbit into
// v45 / v49 / v53 / v57).  After 16 columns the low half of V0 leaves: its ONE bits are banked (the zeros follow from
// the number of half steps), V moves down by v_alignbit and all ones enter on top.  Only for tiles whose lanes all
// have the same whole number of blocks: there is no masked path here.
#define HG_REGS0 "v20", "v21", "v33", "v24", "v25", "v37"
#define HG_REGS1 "v20", "v45", "v49", "v24", "v53", "v57"
#define HG_DRAIN_(B0, B1, I1, I2, V0, P00, P01, V1, P10, P11) \
    TG_Ua("v26", V1, P10, B0) TG_Ub("v26", P11, B1) TG_ADC("v27", "s[48:49]", V1, "v26", I1) TG_N(V1, "v26", "v27")
#define HG_DRAIN(B, I, V0, P00, P01, V1, P10, P11) HG_DRAIN_(B, I, V0, P00, P01, V1, P10, P11)
#define HG_FILL0(s, V0, P00, P01, V1, P10, P11) TG_FILL0(s, DG_X0, DG_CE, V0, P00, P01)
#define HG_ODD(s, V0, P00, P01, V1, P10, P11) TG_FILL1(s, DG_X1, DG_X0, DG_CE, DG_CO, V0, P00, P01, V1, P10, P11)
#define HG_EVEN(s, V0, P00, P01, V1, P10, P11) TG_FILL1(s, DG_X0, DG_X1, DG_CO, DG_CE, V0, P00, P01, V1, P10, P11)
#define HG_HALF_(s0, s1, s2, s3, s4, s5, s6, s7, s8, s9, s10, s11, s12, s13, s14, s15, V0, P00, P01, V1, P10, P11)    \
    ".p2align 3\n\ts_nop 0\n\t"                                                                                         \
    HG_FILL0(s0, V0, P00, P01, V1, P10, P11) HG_ODD(s1, V0, P00, P01, V1, P10, P11) HG_EVEN(s2, V0, P00, P01, V1, P10, P11) \
    HG_ODD(s3, V0, P00, P01, V1, P10, P11) HG_EVEN(s4, V0, P00, P01, V1, P10, P11) HG_ODD(s5, V0, P00, P01, V1, P10, P11)  \
    HG_EVEN(s6, V0, P00, P01, V1, P10, P11) HG_ODD(s7, V0, P00, P01, V1, P10, P11) HG_EVEN(s8, V0, P00, P01, V1, P10, P11) \
    HG_ODD(s9, V0, P00, P01, V1, P10, P11) HG_EVEN(s10, V0, P00, P01, V1, P10, P11) HG_ODD(s11, V0, P00, P01, V1, P10, P11) \
    HG_EVEN(s12, V0, P00, P01, V1, P10, P11) HG_ODD(s13, V0, P00, P01, V1, P10, P11) HG_EVEN(s14, V0, P00, P01, V1, P10, P11) \
    HG_ODD(s15, V0, P00, P01, V1, P10, P11) HG_DRAIN(DG_X1, DG_CO, V0, P00, P01, V1, P10, P11)
#define HG_HALF(s0, s1, s2, s3, s4, s5, s6, s7, s8, s9, s10, s11, s12, s13, s14, s15, R) \
    HG_HALF_(s0, s1, s2, s3, s4, s5, s6, s7, s8, s9, s10, s11, s12, s13, s14, s15, R)
// the low 16 rows leave (ones counted), 16 fresh rows enter
#define HG_DOWN "v_and_b32 v22, 0xffff, v20\n\tv_bcnt_u32_b32 %[bank], v22, %[bank]\n\t" \
                "v_alignbit_b32 v20, v24, v20, 16\n\tv_alignbit_b32 v24, -1, v24, 16\n\t"

// All blocks of a tile.  On entry nxt holds the rows of block 0 and p points at them, V[0], V[1] / P[0], P[1] are the
// limbs the window starts with; `ones` collects the one bits of the rows that leave (32 per block).
__device__ __forceinline__ void diag_halves_dna(uint32_t (&V)[3], uint32_t (&P)[3][2], uint32_t &ones,
                                                const uint32_t *&p, uint32_t (&nxt)[4], int iters)
{
    register uint32_t V0 asm("v20") = V[0], V1 asm("v24") = V[1];
    register uint32_t P00 asm("v21") = P[0][0], P10 asm("v25") = P[1][0], P20 asm("v29");
    register uint32_t P01 asm("v33") = P[0][1], P11 asm("v37") = P[1][1], P21 asm("v41");
    register uint32_t w0 asm("v32"), w1 asm("v36");
    uint32_t n0 = nxt[0], n1 = nxt[1], n2 = nxt[2], n3 = nxt[3], bank = ones;
    int it = 0;
    do {
        asm volatile(
            "v_mov_b32 v29, %[n0]\n\tv_mov_b32 v41, %[n1]\n\tv_mov_b32 v32, %[n2]\n\tv_mov_b32 v36, %[n3]\n\t"
            TG_LOADS(1024, 1280, 1536, 1792)
            HG_HALF(0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, HG_REGS0)
            HG_DOWN
            "v_alignbit_b32 v45, v25, v21, 16\n\tv_alignbit_b32 v49, v37, v33, 16\n\t"
            "v_alignbit_b32 v53, v29, v25, 16\n\tv_alignbit_b32 v57, v41, v37, 16\n\t"
            HG_HALF(16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31, HG_REGS1)
            HG_DOWN
            "v_mov_b32 v21, v25\n\tv_mov_b32 v33, v37\n\tv_mov_b32 v25, v29\n\tv_mov_b32 v37, v41\n\t"
            "s_waitcnt vmcnt(0)"
            : "+v"(V0), "+v"(V1), "+v"(P00), "+v"(P01), "+v"(P10), "+v"(P11), "=&v"(P20), "=&v"(P21),
              [w0] "=&v"(w0), [w1] "=&v"(w1), [bank] "+v"(bank), [n0] "+v"(n0), [n1] "+v"(n1), [n2] "+v"(n2), [n3] "+v"(n3)
            : [np] "v"(p)
            : "v22", "v26", "v23", "v27", "v35", "v47", "v39", "v51", "v45", "v49", "v53", "v57",
              "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49");
        p += 4 * 64;
    } while (++it < iters);
    V[0] = V0; V[1] = V1;
    nxt[0] = n0; nxt[1] = n1; nxt[2] = n2; nxt[3] = n3;
    ones = bank;
}

// ---- two limbs, match masks from LDS -----------------------------------------------------------------
// The plane-folding stream above spends 6 of its 10 instructions per column on U = V & Match: two v_bfe_i32 for the text
// bits and two v_bitop3 per limb.  Here the four match masks of the window (one per symbol, 64 rows) are in LDS and a
// column costs v_bfe_u32 + v_lshl_or_b32 (symbol -> address), one ds_read2st64_b32 (both limbs, LDS pipe) and one
// v_and_b32 per limb: 8 VALU instructions per column, two of them 32-bit encodings, 8.7 with everything around them.
// A lane reads only what it wrote itself (row * 256 + lane * 4), a wave's LDS operations complete in order, so the
// table needs no barrier; 4 KB per wave.  The stream is generated (tools/gen_diag_lds.py -> malcs_diag_lds.inc): all
// blocks of a piece are ONE statement with its own loop, the LDS read of a column is issued six columns ahead (across
// block boundaries), the rows of the next block are loaded by the statement itself (sbase + voff).  The text rows of
// such a piece hold 2-bit symbols (PIECE_CODES, written by k_diag_layout).
// On entry: V[0 ... 1] / P[0 ... 1] the limbs the window starts with, nxt the rows of block 0 (they start at
// rows_u + lane); `ones` collects the one bits of the rows that leave.  The arithmetic is that of diag_halves_dna.
#include "malcs_diag_lds.inc"
// join_slots: the tile's words in the join buffer when the piece is a part of a cut tile (joined), whose other part's
// word comes back in `pre` if it was there DIAG_LDS_JOINPRE blocks before the end (else zero).
__device__ __forceinline__ void diag_halves_lds(uint32_t (&V)[3], uint32_t (&P)[3][2], uint32_t &ones, const uint32_t *rows_u,
                                                unsigned lane, const uint32_t (&nxt)[4], int iters, uint32_t *tab,
                                                const unsigned long long *join_slots, bool joined, unsigned long long &pre)
{
    auto masks = [](uint32_t p0, uint32_t p1, uint32_t (&mk)[4]) {
        mk[0] = ~(p0 | p1); mk[1] = p0 & ~p1; mk[2] = ~p0 & p1; mk[3] = p0 & p1;
    };
    uint32_t m0[4], m1[4];
    masks(P[0][0], P[0][1], m0);
    masks(P[1][0], P[1][1], m1);
#pragma unroll
    for (int k = 0; k < 4; k++) {
        tab[(0 + k) * 64 + lane] = m0[k];                                  // aligned ring, slot 0: w[0]
        tab[(4 + k) * 64 + lane] = m1[k];                                  //               slot 1: w[1]
        tab[(8 + k) * 64 + lane] = (m0[k] >> 16) | (m1[k] << 16);          // shifted ring, slot 0: (w[1]:w[0]) >> 16
    }
    register uint32_t V0 asm(DIAG_LDS_V0) = V[0], V1 asm(DIAG_LDS_V1) = V[1];
    register uint32_t q0 asm("v44") = m1[0], q1 asm("v45") = m1[1], q2 asm("v46") = m1[2], q3 asm("v47") = m1[3];
    register uint32_t p0 asm("v28") = nxt[0], p1 asm("v29") = nxt[1], t0 asm("v32") = nxt[2], t1 asm("v36") = nxt[3];
    // the stream puts a column's symbol into byte 1 of the address: the table must start at LDS address 0 (it is the
    // kernel's only shared array)
    if ((uint32_t)reinterpret_cast<uintptr_t>(tab) != 0u) __builtin_trap();
    register uint32_t base asm("v30") = lane * 4u;
    uint32_t voff = lane * 4u, bank = ones;
    uint32_t cnt = (uint32_t)__builtin_amdgcn_readfirstlane(iters);
    const uint32_t trig = (uint32_t)__builtin_amdgcn_readfirstlane(joined ? DIAG_LDS_JOINPRE : 0);   // cnt is never 0 at the start of a block
    const uint32_t vj = lane * 8u;
    unsigned long long got = 0ull;
    asm volatile(DIAG_LDS_STREAM
                 : "+v"(V0), "+v"(V1), "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3), "+v"(p0), "+v"(p1), "+v"(t0), "+v"(t1),
                   [bank] "+&v"(bank), [voff] "+&v"(voff), [cnt] "+&s"(cnt), "+v"(base),  // early clobber: voff and base start equal
                   [pre] "+&v"(got)
                 : [sbase] "s"(rows_u), [trig] "s"(trig), [vj] "v"(vj), [sj] "s"(join_slots)
                 : "v22", "v23", "v26", "v27", "v31", "v35", "v33", "v37", "v38", "v39", "v53", "v54", "v40", "v41", "v42", "v43", "v48", "v49", "v50", "v51",
                   "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79",
                   
                   "s40", "s41", "s42", "s44", "s45", "s48", "s49", "scc", "memory");
    V[0] = V0; V[1] = V1;
    ones = bank;
    pre = got;
}

// ---- three limbs, match masks from LDS ------------------------------------------------------------------------
// DIAG_LDS3_STREAM (tools/gen_diag_lds.py): `blocks` plain blocks of a three-limb piece whose text rows hold 2-bit
// symbols (PIECE_CODES), 10.7 VALU instructions per column where the plane-folding statements above have 14.5.  On
// entry V[0 ... 1] / P[0 ... 1] are the limbs the window starts with and nxt the rows of its first block (at
// rows_b + lane); on return they are the limbs it ends with and the rows of the block behind the last one done.
// `ones` collects the one bits of the limbs that leave (one per block).  The table is the two-limb stream's (16 rows
// of 64 dwords at LDS address 0): limb j in slot j % 4.
__device__ __forceinline__ void diag_thirds_lds(uint32_t (&V)[3], uint32_t (&P)[3][2], uint32_t &ones, const uint32_t *rows_b,
                                                unsigned lane, uint32_t (&nxt)[4], int blocks, uint32_t *tab)
{
    auto masks = [](uint32_t p0, uint32_t p1, uint32_t (&mk)[4]) {
        mk[0] = ~(p0 | p1); mk[1] = p0 & ~p1; mk[2] = ~p0 & p1; mk[3] = p0 & p1;
    };
    uint32_t m0[4], m1[4], m2[4];
    masks(P[0][0], P[0][1], m0);
    masks(P[1][0], P[1][1], m1);
    masks(nxt[0], nxt[1], m2);                                             // the rows of a block start with the planes of the limb that enters there
#pragma unroll
    for (int k = 0; k < 4; k++) {
        tab[(0 + k) * 64 + lane] = m0[k];
        tab[(4 + k) * 64 + lane] = m1[k];
        tab[(8 + k) * 64 + lane] = m2[k];
    }
    if ((uint32_t)reinterpret_cast<uintptr_t>(tab) != 0u) __builtin_trap();   // (the symbol goes into byte 1 of the address)
    register uint32_t V0 asm("v20") = V[0], V1 asm("v21") = V[1], V2 asm("v24") = ~0u;
    register uint32_t a0 asm("v57") = P[0][0], a1 asm("v58") = P[0][1], b0 asm("v59") = P[1][0], b1 asm("v60") = P[1][1];
    register uint32_t c0 asm("v61") = nxt[0], c1 asm("v62") = nxt[1];          // the planes of the window's three limbs follow it
    register uint32_t p0 asm("v28") = 0u, p1 asm("v29") = 0u;
    register uint32_t t00 asm("v32") = nxt[2], t01 asm("v36") = nxt[3], t10 asm("v33") = 0u, t11 asm("v37") = 0u;
    register uint32_t base asm("v30") = lane * 4u;
    uint32_t voff = lane * 4u, bank = ones;
    uint32_t cnt = (uint32_t)__builtin_amdgcn_readfirstlane(blocks);
    asm volatile(DIAG_LDS3_STREAM
                 : "+v"(V0), "+v"(V1), "+v"(V2), "+v"(a0), "+v"(a1), "+v"(b0), "+v"(b1), "+v"(c0), "+v"(c1), "+v"(p0), "+v"(p1),
                   "+v"(t00), "+v"(t01), "+v"(t10), "+v"(t11), [bank] "+&v"(bank), [voff] "+&v"(voff), [cnt] "+&s"(cnt), "+v"(base)
                 : [sbase] "s"(rows_b)
                 : "v22", "v23", "v26", "v27", "v55", "v56", "v31", "v35", "v38", "v39", "v53", "v54", "v40", "v41", "v42", "v43",
                   "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79",
                   "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95",
                   "s40", "s41", "s42", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "scc", "memory");
    V[0] = V0; V[1] = V1;
    P[0][0] = a0; P[0][1] = a1; P[1][0] = b0; P[1][1] = b1;
    nxt[0] = p0; nxt[1] = p1;
    nxt[2] = (blocks & 1) ? t10 : t00; nxt[3] = (blocks & 1) ? t11 : t01;   // block i leaves the next one's text in the registers of parity i + 1
    ones = bank;
}

// The same for the other classes: loads the compiler does not see, and a wait statement that every
// destination passes through before its first use (the loads are unconditional: the buffer is padded).
template <int J>
__device__ __forceinline__ void row_load_async(const uint32_t *p, uint32_t &dst)
{
    asm volatile("global_load_dword %0, %1, off offset:%2" : "=v"(dst) : "v"(p), "n"(J * 256));
}
template <int N, int... J>
__device__ __forceinline__ void rows_load_async(const uint32_t *p, uint32_t (&dst)[N], std::integer_sequence<int, J...>)
{
    (row_load_async<J>(p, dst[J]), ...);
}
template <int N>
__device__ __forceinline__ void rows_wait(uint32_t (&dst)[N])
{
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(dst[0]));
#pragma unroll
    for (int j = 1; j < N; j++) asm volatile("" : "+v"(dst[j]));
}

template <int NLW, int NP, bool MASKED, int... S>
__device__ __forceinline__ void diag_block(uint32_t (&V)[NLW], const uint32_t (&P)[NLW][NP], const uint32_t (&w)[NP],
                                           uint32_t vm, std::integer_sequence<int, S...>)
{
    diag_state<NLW, NP> st;
    (diag_step<NLW, NP, MASKED, S>(V, P, w, vm, st), ...);
}

// ---- the window kernel -------------------------------------------------------------------------
// One workgroup (one wave) per piece.  A piece is a whole tile, or one of the two parts of a tile in two-limb mode that
// the plan cut at a block boundary: the front part runs the pair's first columns forwards, the back part runs the
// reversed pair from its far end, and whichever finishes second joins them.  Joining at column J: the front part knows
// Lf(i) = LLCS(pattern[0, i), text[0, J)) for the 64 rows i of its window, the back part Lb(i) = LLCS(pattern[i, m),
// text[J, n)) for the 64 rows of its own; both windows hold the band of diagonals an alignment of cost <= T can visit,
// so LLCS' = max over the rows i in both windows of Lf(i) + Lb(i) is exact under the same condition as a whole tile
// (D' <= T).  In window coordinates x = i - (first row of the front window) the two windows overlap in x = 0 ... S,
// S = 2 kmax - delta (47 or 48), Lf = banked zeros + zero bits of the front V below x, Lb likewise below S - x.
template <int NLW, int NP, bool FUSED = true>
__global__ void __launch_bounds__(64)
k_llcs_diag(const uint32_t *__restrict__ rows, const uint32_t *__restrict__ piece_off,
            const uint32_t *__restrict__ piece_info, const uint32_t *__restrict__ piece_join, int32_t *__restrict__ dist,
            uint32_t *__restrict__ retry_list, const uint32_t *__restrict__ retry_first,
            const uint32_t *__restrict__ retry_cap, uint32_t *__restrict__ retry_count,
            uint32_t *__restrict__ retry_count_next, uint32_t n_classes,
            uint32_t *__restrict__ join_buf, unsigned long long *__restrict__ stamps, uint32_t pad)
{
    constexpr int T = 32 * NLW - 32;
    // dev (MALCS_DIAG_STAMPS): the 100 MHz wall clock at four points of every wave + where it runs
    auto stamp = [&](int k) {
        if (stamps > reinterpret_cast<unsigned long long *>(3) && threadIdx.x == 0) {     // (1, 2: switches, see malcs_diag_launch)
            unsigned long long t = wall_clock64();
            if (k == 0) { uint32_t hw, xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw)); asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc)); stamps[(blockIdx.x - pad) * 6ull + 4] = hw; stamps[(blockIdx.x - pad) * 6ull + 5] = (xcc & 15u) | ((unsigned long long)piece_info[blockIdx.x - pad] << 8); }
            stamps[(blockIdx.x - pad) * 6ull + k] = t;
        }
    };
    if (blockIdx.x < pad) return;                      // (see malcs_diag_launch: workgroups that only take the first turn of the hardware's round)
    stamp(0);
    // the counters of the NEXT launch start at zero: nothing in flight reads or writes that set now
    if (blockIdx.x == pad && threadIdx.x < n_classes) retry_count_next[threadIdx.x] = 0u;
    const unsigned lane = threadIdx.x;
    const uint32_t tile = blockIdx.x - pad;
    // where the piece's rows are and what it is: two scalar loads side by side (the compiler makes them vector loads,
    // at twice the latency, and everything a wave does waits for them)
    // (the slot of the join buffer too: read after the loop, it was two more trips to memory in a row -- the kernel
    // argument, then the table -- at the end of every wave of a cut tile, before the swap could even leave)
    uint32_t off_s, info, jt;
    {
        const uint32_t t4 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tile * 4u));
        asm volatile("s_load_dword %0, %3, %6\n\ts_load_dword %1, %4, %6\n\ts_load_dword %2, %5, %6\n\ts_waitcnt lgkmcnt(0)"
                     : "=&s"(off_s), "=&s"(info), "=&s"(jt) : "s"(piece_off), "s"(piece_info), "s"(piece_join), "s"(t4));
    }
    if (!info) return;                                 // the grid is an upper bound of the plan's pieces (decided on the device)
    // the tile's slots in the join buffer (one 64-bit word per lane); for a piece that is not joined: the first tile's, unused
    const unsigned long long *const join_slots = reinterpret_cast<const unsigned long long *>(join_buf) + (size_t)((info & PIECE_JOINED) ? jt : 0u) * (JOIN_WORDS * 64u);
    const uint32_t *rows_u = rows + (size_t)off_s * 64u;                                               // the piece's rows
    const uint32_t *p = rows_u + lane;
    const int iters = (int)(info & PIECE_BLOCKS);
    const bool halves = NLW == 3 && NP == 2 && FUSED && (info & PIECE_HALF);     // two limbs, half-limb steps (T = 48)
    // the rows of block 0 are asked for first, together with everything else a wave starts with: one trip to memory
    uint32_t nxt[2 * NP];
    rows_load_async(p + (META_ROWS + (NLW - 1) * NP) * 64, nxt, std::make_integer_sequence<int, 2 * NP>());
    const uint32_t pid = p[0], m = p[64], n = p[128], kc = p[192];
    const int kmax = (int)(kc & 0xffffu);          // block `it` = columns [32 it, 32 it + 32), its limb k = rows from 32 (it + k) - kmax
    p += META_ROWS * 64;

    auto rowmask = [&](int rs) -> uint32_t {       // of the 32 rows from rs, those inside the matrix
        const int vb = (int)m - rs;
        const uint32_t below = vb >= 32 ? ~0u : (vb <= 0 ? 0u : ((1u << vb) - 1u));
        const uint32_t above = rs >= 0 ? ~0u : (rs <= -32 ? 0u : (~0u << -rs));
        return below & above;
    };

    uint32_t P[NLW][NP], V[NLW];
#pragma unroll
    for (int k = 0; k < NLW - 1; k++) {
#pragma unroll
        for (int q = 0; q < NP; q++) P[k][q] = p[(k * NP + q) * 64];
        const int rs = 32 * k - kmax;              // rows above the matrix: V = 0 neither matches nor carries
        V[k] = rs >= 0 ? ~0u : (rs <= -32 ? 0u : (~0u << -rs));
    }
    p += (NLW - 1) * NP * 64;
    // the join's table (see there): entry (front nibble | back nibble << 4) = sum << 16 | largest prefix sum of the four
    // steps front bit - back bit, lowest bit first.  Four entries per lane, worked out HERE, while the wave's first rows
    // are on their way (behind the wait below it was 0.2 us of every wave's start); a wave's LDS operations complete in
    // order, so no barrier
    __shared__ uint32_t join_tab[NLW == 3 && NP == 2 && FUSED ? 256 : 1];
    if constexpr (NLW == 3 && NP == 2 && FUSED) {
        if (info & PIECE_JOINED) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint32_t idx = lane * 4u + (uint32_t)j;
                int t = 0, mp = 0;
#pragma unroll
                for (int k = 0; k < 4; k++) { t += (int)((idx >> k) & 1u) - (int)((idx >> (4 + k)) & 1u); mp = max(mp, t); }
                join_tab[idx] = ((uint32_t)t << 16) | (uint32_t)mp;
            }
        }
    }
    rows_wait(nxt);

    uint32_t banked = 0;
    stamp(1);

    // blocks [0, plain_hi) of a lane need no masks: all 32 columns exist and the leaving limb has no rows beyond the
    // matrix.  Rows above the matrix (V = 0 from the start, so they read as zero bits) all leave with the first three
    // limbs: when those are banked without a mask, `above` takes them out again.
    int plain_hi = 0x7fffffff, above = 0;
    if (pid != 0xffffffffu) {
        plain_hi = (int)min(n >> 5, (m + (uint32_t)kmax) >> 5);
    }

    int32_t d = 0;
    bool have_d = false, skip = false;                 // skip: the other part of the tile reports this lane's pair
    __shared__ __attribute__((aligned(4096))) uint32_t lds_masks[NLW == 3 && NP == 2 && FUSED ? 16 * 64 : 1];   // the match masks of the LDS streams
    if constexpr (NLW == 3 && NP == 2 && FUSED) {
        if (halves) {                                  // every block of every lane is plain (the plan saw to that)
            uint32_t ones = 0;
            unsigned long long pre = 0ull;             // the other part's join word, if it was there shortly before the end
            if (info & PIECE_CODES) {
                diag_halves_lds(V, P, ones, rows_u + (META_ROWS + (NLW - 1) * NP) * 64, lane, nxt, iters, lds_masks,
                                join_slots, (info & PIECE_JOINED) != 0u, pre);
            } else {
                diag_halves_dna(V, P, ones, p, nxt, iters);
            }
            stamp(2);
            if (info & PIECE_JOINED) {
                // the window now holds rows [32 iters - kmax, + 64) (inside the matrix: the plan leaves two blocks on
                // either side of a cut); `ones` = one bits of the rows that left.  The join needs 48 of the 64 rows and
                // the count, which fit one 64-bit word: every lane SWAPS its word into the tile's slot (agent scope: one
                // trip to memory, coherent across the XCDs' L2s by itself).  Zero comes back to the part that finishes
                // first, which is done; the other part gets the first one's word, joins, and clears the slot for the
                // next launch.  (Round 2 began with three stores, a wait, an atomic add on a flag per tile and three
                // loads: three round trips in a row at the end of every launch; a __threadfence() in place of the wait
                // writes back and invalidates the whole L2 per wave, 35 -> 103 us.)
                const bool back = info & PIECE_BACK;
                skip = true;
                if (pid != 0xffffffffu) {
                    unsigned long long *slot = const_cast<unsigned long long *>(join_slots) + lane;
                    const uint64_t mv = (((uint64_t)V[1] << 32) | V[0]) & 0xffffffffffffull;
                    // (per lane: the other part's swap is one instruction, but its 64 words need not land together)
                    uint64_t theirs = pre;
                    if (theirs == 0ull)
                        theirs = __hip_atomic_exchange(slot, mv | ((uint64_t)(ones + 1u) << 48), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (theirs != 0ull) {
                        __hip_atomic_store(slot, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        const uint64_t ov = theirs & 0xffffffffffffull;
                        const uint32_t oo = (uint32_t)(theirs >> 48) - 1u;
                        const uint64_t Vf = back ? ov : mv, Vb = back ? mv : ov;
                        const int S = 2 * kmax - (int)(n - m);                       // 47 or 48
                        const uint64_t A = ~Vf, B = ~Vb & ((1ull << S) - 1ull);
                        const uint64_t Br = __brevll(B) >> (64 - S);                 // bit x: row S - 1 - x of the back window
                        const uint64_t Am = A & ((1ull << S) - 1ull);                // rows beyond S: no step
                        const uint32_t a0 = (uint32_t)Am, a1 = (uint32_t)(Am >> 32), b0 = (uint32_t)Br, b1 = (uint32_t)(Br >> 32);
                        // x = 0: nothing of the front window, S rows of the back one; step x moves row x from the back
                        // window's side to the front one's: a walk of S steps of +1 / 0 / -1 whose largest prefix sum is
                        // asked for.  Four steps at a time through join_tab (index: four bits of either side): the wave
                        // is alone on its SIMD by now, where an instruction issues every five cycles or so, and a step
                        // at a time the walk was 260 of them (0.7 us at the end of every launch).
                        const uint32_t x0 = (a0 & 0x0f0f0f0fu) | ((b0 << 4) & 0xf0f0f0f0u);          // byte k: rows 8 k ... 8 k + 3
                        const uint32_t y0 = ((a0 >> 4) & 0x0f0f0f0fu) | (b0 & 0xf0f0f0f0u);          //         rows 8 k + 4 ... 8 k + 7
                        const uint32_t x1 = (a1 & 0x0f0fu) | ((b1 << 4) & 0xf0f0u);
                        const uint32_t y1 = ((a1 >> 4) & 0x0f0fu) | (b1 & 0xf0f0u);
                        uint32_t e[MALCS_DIAG_HALF_T / 4];
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            e[2 * k] = join_tab[(x0 >> (8 * k)) & 0xffu];
                            e[2 * k + 1] = join_tab[(y0 >> (8 * k)) & 0xffu];
                        }
#pragma unroll
                        for (int k = 0; k < MALCS_DIAG_HALF_T / 8 - 4; k++) {
                            e[8 + 2 * k] = join_tab[(x1 >> (8 * k)) & 0xffu];
                            e[9 + 2 * k] = join_tab[(y1 >> (8 * k)) & 0xffu];
                        }
                        int t = 0, best = 0;
#pragma unroll
                        for (int k = 0; k < MALCS_DIAG_HALF_T / 4; k++) {
                            best = max(best, t + (int)(e[k] & 0xffffu));
                            t += (int)e[k] >> 16;
                        }
                        best += __popcll(B);
                        // zero bits of the rows that left either window (all inside the matrix but the kmax rows above it)
                        const uint32_t zeros = n - ones - oo - 2u * (uint32_t)kmax + (uint32_t)best;
                        d = (int32_t)(m + n - 2u * zeros);
                        have_d = true;
                        skip = false;
                    }
                }
            } else {
                banked = 32u * (uint32_t)iters - ones;
                above = kmax;                          // rows above the matrix were banked as zeros
            }
        }
    }
    if (!have_d && !(halves && (info & PIECE_JOINED))) {
        // text rows that hold 2-bit symbols (PIECE_CODES: written for the LDS streams) as the two planes the other paths read
        auto planes_of = [](uint32_t c0, uint32_t c1, uint32_t (&w)[NP]) {
            auto even = [](uint32_t x) -> uint32_t {
                x &= 0x55555555u; x = (x | (x >> 1)) & 0x33333333u; x = (x | (x >> 2)) & 0x0f0f0f0fu;
                x = (x | (x >> 4)) & 0x00ff00ffu; x = (x | (x >> 8)) & 0xffffu;
                return x;
            };
            w[0] = even(c0) | (even(c1) << 16);
            if constexpr (NP >= 2) w[1] = even(c0 >> 1) | (even(c1 >> 1) << 16);
        };
        const bool codes = NP == 2 && !halves && (info & PIECE_CODES);
        for (int it = halves ? iters : 0; it < iters;) {
            if constexpr (NLW == 3 && NP == 2 && FUSED) {
                if (codes && it == 0) {
                    // the blocks every lane of the wave can run without masks, through the LDS stream (all of them in a
                    // piece the plan made: it grows the sequences to whole blocks and to the tile's longest)
                    int lo = min(plain_hi, iters);
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) lo = min(lo, __shfl_xor(lo, o, 64));
                    int n_lds = __builtin_amdgcn_readfirstlane(lo);
                    if (stamps == reinterpret_cast<unsigned long long *>(3)) n_lds = max(1, n_lds / 2);    // (test switch: the rest through the plane paths)
                    if (n_lds >= 1 && stamps != reinterpret_cast<unsigned long long *>(2)) {
                        uint32_t ones = 0;
                        diag_thirds_lds(V, P, ones, rows_u + (META_ROWS + (NLW - 1) * NP) * 64, lane, nxt, n_lds, lds_masks);
                        banked += 32u * (uint32_t)n_lds - ones;
                        above = min(kmax, 32 * n_lds);         // the rows above the matrix in the limbs that left, banked as zeros (V = 0 there)
                        p += (size_t)n_lds * 2 * NP * 64;
                        it = n_lds;
                        continue;
                    }
                }
                auto plain3 = [&](int i) { return !codes && i + 3 <= iters && __builtin_amdgcn_ballot_w64(i + 3 > plain_hi) == 0; };
                if (plain3(it)) {
                    if (it == 0) above = kmax;                 // kmax <= 64: rows [-kmax, 0) lie in the first three leaving limbs
                    diag_triples_dna3(V, P, banked, p, nxt, it, plain3);
                    continue;
                }
            }
            uint32_t w[NP];
#pragma unroll
            for (int q = 0; q < NP; q++) { P[NLW - 1][q] = nxt[q]; w[q] = nxt[NP + q]; }
            if constexpr (NP == 2) { if (codes) planes_of(nxt[2], nxt[3], w); }
            V[NLW - 1] = ~0u;
            p += 2 * NP * 64;                          // rows of the next block (the buffer ends with one block of padding)
            const int hi = (int)n - 32 * it;           // one past the last valid step of the block: steps beyond must not change V
            const uint32_t vm = hi >= 32 ? ~0u : (hi <= 0 ? 0u : ((1u << hi) - 1u));
            const bool fast = __builtin_amdgcn_ballot_w64(vm != ~0u) == 0;
            if constexpr (NLW == 3 && NP == 2 && FUSED) {
                if (fast) {
                    diag_block_dna3(V, P, w, p, nxt);
                } else {                               // last blocks of a pair
                    rows_load_async(p, nxt, std::make_integer_sequence<int, 2 * NP>());
                    diag_block<NLW, NP, true>(V, P, w, vm, std::make_integer_sequence<int, 32 + NLW - 1>());
                    rows_wait(nxt);
                }
            } else {
                rows_load_async(p, nxt, std::make_integer_sequence<int, 2 * NP>());
                if (fast) diag_block<NLW, NP, false>(V, P, w, vm, std::make_integer_sequence<int, 32 + NLW - 1>());
                else      diag_block<NLW, NP, true>(V, P, w, vm, std::make_integer_sequence<int, 32 + NLW - 1>());
                rows_wait(nxt);
            }
            // the window moves down one limb
            banked += __popc(~V[0] & rowmask(32 * it - kmax));
#pragma unroll
            for (int k = 0; k < NLW - 1; k++) {
                V[k] = V[k + 1];
#pragma unroll
                for (int q = 0; q < NP; q++) P[k][q] = P[k + 1][q];
            }
            it++;
        }
        bool joined3 = false;
        if constexpr (NLW == 3 && NP == 2 && FUSED) {
            if (info & PIECE_JOINED) {
                // A part of a cut tile in three-limb mode.  As in two-limb mode (above) the window now holds rows
                // [32 iters - kmax, + 96), all inside the matrix, and the join needs the first S = 2 kmax - delta (63 or
                // 64) of them and the zeros that left: 64 + 17 bits, two words.  The rows go to the part's own word of
                // the tile's slot first and are WAITED for (an agent-scope store is at the memory side when it is
                // acknowledged), then the count is swapped in as in two-limb mode: whoever gets the other part's count
                // back finds its rows there, whatever the order in which the two parts' words travelled.
                joined3 = true;
                skip = true;
                const bool back = info & PIECE_BACK;
                if (pid != 0xffffffffu) {
                    unsigned long long *slot = const_cast<unsigned long long *>(join_slots) + lane;
                    const uint64_t mv = ((uint64_t)V[1] << 32) | V[0];
                    __hip_atomic_store(slot + (back ? 128 : 64), mv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    const uint32_t left = banked - (uint32_t)above;          // zero bits of the rows that left, inside the matrix
                    const uint64_t theirs = __hip_atomic_exchange(slot, (uint64_t)left + 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (theirs != 0ull) {
                        const uint64_t ov = __hip_atomic_load(slot + (back ? 64 : 128), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(slot, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        const uint32_t oleft = (uint32_t)theirs - 1u;
                        const uint64_t Vf = back ? ov : mv, Vb = back ? mv : ov;
                        const int S = 2 * kmax - (int)(n - m);                       // 63 or 64
                        const uint64_t ms = S >= 64 ? ~0ull : ((1ull << S) - 1ull);
                        const uint64_t Am = ~Vf & ms, B = ~Vb & ms;
                        const uint64_t Br = __brevll(B) >> (64 - S);                 // bit x: row S - 1 - x of the back window
                        const uint32_t a0 = (uint32_t)Am, a1 = (uint32_t)(Am >> 32), b0 = (uint32_t)Br, b1 = (uint32_t)(Br >> 32);
                        const uint32_t x0 = (a0 & 0x0f0f0f0fu) | ((b0 << 4) & 0xf0f0f0f0u), y0 = ((a0 >> 4) & 0x0f0f0f0fu) | (b0 & 0xf0f0f0f0u);
                        const uint32_t x1 = (a1 & 0x0f0f0f0fu) | ((b1 << 4) & 0xf0f0f0f0u), y1 = ((a1 >> 4) & 0x0f0f0f0fu) | (b1 & 0xf0f0f0f0u);
                        int t = 0, best = 0;
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            const uint32_t e0 = join_tab[(x0 >> (8 * k)) & 0xffu], e1 = join_tab[(y0 >> (8 * k)) & 0xffu];
                            best = max(best, t + (int)(e0 & 0xffffu)); t += (int)e0 >> 16;
                            best = max(best, t + (int)(e1 & 0xffffu)); t += (int)e1 >> 16;
                        }
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            const uint32_t e0 = join_tab[(x1 >> (8 * k)) & 0xffu], e1 = join_tab[(y1 >> (8 * k)) & 0xffu];
                            best = max(best, t + (int)(e0 & 0xffffu)); t += (int)e0 >> 16;
                            best = max(best, t + (int)(e1 & 0xffffu)); t += (int)e1 >> 16;
                        }
                        best += __popcll(B);
                        const uint32_t zeros = left + oleft + (uint32_t)best;
                        d = (int32_t)(m + n - 2u * zeros);
                        skip = false;
                    }
                }
            }
        }
        if (!joined3) {
            uint32_t zeros = banked - (uint32_t)above;
#pragma unroll
            for (int k = 0; k < NLW - 1; k++) zeros += __popc(~V[k] & rowmask(32 * (iters + k) - kmax));
            d = (int32_t)(m + n - 2u * zeros);
        }
    }

    stamp(3);
    if (pid != 0xffffffffu && !skip) {
        if (d <= (halves ? MALCS_DIAG_HALF_T : T)) {
            // scattered by pair number: a line per lane.  Written through, the lines leave during the launch; left dirty
            // in the L2s they are written back when the kernel ends, between this launch and the next
            __builtin_nontemporal_store(d, &dist[pid]);
        } else {
            const uint32_t cls = kc >> 16;
#ifdef MALCS_DEV
            if (stamps == reinterpret_cast<unsigned long long *>(1)) return;      // dev build only (MALCS_DIAG_NORETRY): timing experiments with wrong results
#endif
            const uint32_t idx = atomicAdd(&retry_count[cls], 1u);
            if (idx < retry_cap[cls]) retry_list[retry_first[cls] + idx] = pid;
        }
    }
}

typedef void (*diag_kernel_t)(const uint32_t *, const uint32_t *, const uint32_t *, const uint32_t *, int32_t *, uint32_t *,
                              const uint32_t *, const uint32_t *, uint32_t *, uint32_t *, uint32_t, uint32_t *, unsigned long long *, uint32_t);
template <int NP>
diag_kernel_t pick_diag(int nlw)
{
    switch (nlw) {
    case 2: return k_llcs_diag<2, NP>;
    case 3: if (getenv("MALCS_DIAG_GENERIC")) return k_llcs_diag<3, NP, false>; else return k_llcs_diag<3, NP>;   // dev switch: A/B the fused block
    case 4: return k_llcs_diag<4, NP>;
    case 5: return k_llcs_diag<5, NP>;
    case 8: return k_llcs_diag<8, NP>;
    default: return nullptr;
    }
}

}  // namespace

int malcs_diag_threshold(int nlw) { return 32 * nlw - 32; }

void malcs_diag_release(malcs_batch *b)
{
    diag_plan &d = b->diag;
    if (d.d_rows) malcs_pool_free(b->ctx, d.d_rows);          // the other arrays of the plan are parts of this allocation
    d = diag_plan();
}

// Everything the window class of a batch needs before it can run, in two parts.
//
// malcs_diag_setup, once per batch, from the LENGTHS of its pairs alone (which malcs_batch_create is given): the pairs of
// the class and their retry classes in device memory, one allocation for everything a plan writes (sized by upper
// bounds: the row buffer holds the blocks of the untrimmed texts, which the trimmed ones cannot exceed), the sub-pairs of
// the sample.
//
// malcs_diag_plan, per plan, everything that depends on what the sequences HOLD, queued on the context's stream with
// nothing coming back to the host:
//   1. k_plan_ends: common ends, what is left of every pair, histogram of block counts; in the same launch the
//      sample: about a thousand pairs, each cut in the middle, through the full-matrix wave;
//   2. k_plan_scatter puts the pairs into their slots (counting sort);
//   3. k_diag_layout writes the rows of every piece; its first workgroup deals the pieces to the SIMDs: the table of the launch.
// (malcs_plan.inc has the kernels and why there are three launches and no more.)
// The launch that follows is ordered behind the plan by the stream.
int malcs_diag_setup(malcs_batch *b, int nlw, const std::vector<uint32_t> &pairs, const std::vector<uint8_t> &retry_class,
                     const std::vector<diag_retry> &classes)
{
    malcs_diag_release(b);
    diag_plan &d = b->diag;
    if (pairs.empty()) return MALCS_OK;
    if (b->n_planes > 4 || !(nlw == 2 || nlw == 3 || nlw == 4 || nlw == 5 || nlw == 8)) {
        malcs_set_error("no window kernel for %d limbs and %d bit planes", nlw, b->n_planes);
        return MALCS_ERR_INTERNAL;
    }
    malcs_ctx *ctx = b->ctx;
    hipStream_t st = ctx->stream;
    const uint32_t count = (uint32_t)pairs.size();
    const uint32_t NP = (uint32_t)b->n_planes;
    // (every pair here passed malcs_dist_plan's test: length difference <= T, at least 2 nlw + 2 blocks; it also noted the
    // longest sequence among them and the blocks of their texts)
    const uint32_t max_len = b->plan_window_longest;
    if (!max_len) { malcs_set_error("window plan: empty sequences"); return MALCS_ERR_INTERNAL; }
    const uint32_t HB = (max_len + 31u) / 32u + 2u;          // histogram bins per group: block counts 0 ... HB - 1
    if (2u * HB > DEAL_HB2) { malcs_set_error("window plan: sequences of %u symbols are beyond its histogram", max_len); return MALCS_ERR_TOO_LONG; }
    const uint32_t n_tiles = (count + 63u) / 64u;
    const uint32_t S = (uint32_t)std::max(1, ctx->prop.multiProcessorCount) * 4u;
    const bool can_half = nlw == 3 && NP == 2 && !getenv("MALCS_DIAG_GENERIC");
    int cut_mode = -1;                                       // MALCS_DIAG_CUT=0 / 1: never / always
    if (const char *env = getenv("MALCS_DIAG_CUT")) if (*env) cut_mode = atoi(env) != 0;
    const bool cut_possible = can_half && (cut_mode > 0 || (cut_mode < 0 && 2u * n_tiles <= 4u * S));
    const uint32_t n_launch = cut_possible ? 2u * n_tiles : n_tiles;

    // the sample: pairs at equal distances through the class, each as two sub-pairs (cut at the block nearest the
    // middle of either sequence); all sub-pairs in ONE full-matrix class, the narrowest their longest allows -- the
    // sample is a few hundred waves that run side by side, so its time is the time of one wave: columns x limbs
    // The sample: pairs at equal distances through the class, each as SAMPLE_PARTS sub-pairs (cut at block boundaries of
    // the arena: a at the boundaries nearest the equal parts of its length, b at the boundaries nearest the same fractions
    // of its own) in SAMPLE_SLOTS places (the places of a pair sit side by side in a wave; places beyond the parts are empty).  A cut is up
    // to 16 symbols off the alignment's own, which the parts on either side pay for with as many insertions or deletions:
    // known here, taken off again on the device.  All sub-pairs run in ONE full-matrix class, the narrowest their longest
    // allows -- the sample is a few hundred waves that run side by side, so its time is the time of one wave: columns x limbs.
    std::vector<uint32_t> sample;                            // a_off, a_len, b_off, b_len, list (SAMPLE_SLOTS n_sample each); n_sample corrections
    uint32_t n_sample = 0, longest_whole = 0;
    std::vector<uint32_t> sample_tab;                        // the same pairs whole {a_off, a_len, b_off, b_len} for the one-lane sample, which cuts them on the device
    int sW = 1, sL = 1;
    const char *env_half = getenv("MALCS_DIAG_HALF");
    const bool forced = env_half && *env_half;
    if (can_half && !forced && count >= 4096) {
        const size_t stride = count / 1024;
        n_sample = (uint32_t)((count - stride / 2 + stride - 1) / stride);
        const size_t n4 = (size_t)SAMPLE_SLOTS * n_sample;
        sample.assign(5 * n4 + n_sample, 0u);
        uint32_t *sa_off = sample.data(), *sa_len = sa_off + n4, *sb_off = sa_len + n4, *sb_len = sb_off + n4, *list = sb_len + n4, *corr = list + n4;
        uint32_t longest = 0, k = 0;
        for (size_t i = stride / 2; i < count && k < n_sample; i += stride, k++) {
            const uint32_t pid = pairs[i];
            const uint32_t la = b->a_len[pid], lb = b->b_len[pid];
            sample_tab.insert(sample_tab.end(), {b->a_off[pid], la, b->b_off[pid], lb});
            longest_whole = std::max(longest_whole, std::max(la, lb));
            uint32_t ca = 0, cb = 0;                             // where the part starts
            for (uint32_t part = 0; part < SAMPLE_PARTS; part++) {
                uint32_t ea = la, eb = lb;                       // ... and ends
                if (part + 1 < SAMPLE_PARTS) {
                    ea = std::max(ca, 32u * (uint32_t)(((uint64_t)la * (part + 1) / SAMPLE_PARTS + 16u) / 32u));
                    ea = std::min(ea, 32u * (la / 32u));
                    const uint32_t ideal = (uint32_t)((uint64_t)ea * lb / std::max(la, 1u));
                    eb = std::min(std::max(cb, 32u * ((ideal + 16u) / 32u)), 32u * (lb / 32u));
                    corr[k] += 2u * (eb > ideal ? eb - ideal : ideal - eb);
                }
                const size_t at = (size_t)SAMPLE_SLOTS * k + part;
                sa_off[at] = b->a_off[pid] + (ca / 32u) * NP; sa_len[at] = ea - ca;
                sb_off[at] = b->b_off[pid] + (cb / 32u) * NP; sb_len[at] = eb - cb;
                longest = std::max(longest, std::min(ea - ca, eb - cb));
                ca = ea; cb = eb;
            }
            for (uint32_t slot = 0; slot < SAMPLE_SLOTS; slot++) list[(size_t)SAMPLE_SLOTS * k + slot] = SAMPLE_SLOTS * k + slot;   // (slots beyond the parts: empty sequences)
        }
        n_sample = k;
        const uint32_t words = std::max(1u, (longest + 63u) / 64u);
        while ((uint32_t)sW * 64u < words) sW *= 2;
        while ((uint32_t)(sW * sL) < words) sL *= 2;
    }

    // retry bookkeeping: one list segment per full-matrix class, sized for the worst case
    d.retry = classes;
    std::vector<uint32_t> cap(classes.size(), 0), first(classes.size(), 0);
    for (uint32_t i = 0; i < count; i++) cap[retry_class[i]]++;
    uint32_t acc = 0;
    for (size_t c = 0; c < classes.size(); c++) { first[c] = acc; acc += cap[c]; d.retry[c].first = first[c]; d.retry[c].cap = cap[c]; }
    const size_t ncls = std::max<size_t>(classes.size(), 1);

    // The row buffer.  Tiles are 64 consecutive slots of the order sorted by (trimmed) block count, a tile's pieces hold
    // the blocks of its longest pair: over one sorted run that sums to at most (all blocks) / 64 + the longest, the
    // two-limb mode has two runs and a tile across their border, and no trimmed text has more blocks than the text itself.
    const uint64_t blocks_cap = (b->plan_window_blocks + (uint64_t)count) / 64u + 4u * (uint64_t)HB + 64u;   // (+ count: a text grows to whole blocks)
    const uint64_t rows_cap = (uint64_t)n_launch * ((uint64_t)META_ROWS + (uint64_t)(nlw - 1) * NP) + blocks_cap * 2u * NP + 2u * NP;
    if (rows_cap >= 0xffffffffull) { malcs_set_error("window class too large; split the batch"); return MALCS_ERR_TOO_LONG; }

    size_t keep = 0;
    auto carve = [](size_t &total, size_t dwords) { const size_t at = total; total += (dwords + 63) & ~(size_t)63; return at; };
    const uint32_t set_words = 4u * HB + 8u;                               // bins and cursors of both groups, two 64-bit column sums, two of the sample
    const size_t at_rows = carve(keep, (size_t)rows_cap * 64u);            // (includes one block of padding: the kernel prefetches unconditionally)
    const size_t at_pairs = carve(keep, count), at_cls = carve(keep, (count + 3) / 4), at_item = carve(keep, (size_t)ITEM_STRIDE * count);
    const size_t at_sets = carve(keep, 2 * (size_t)set_words);
    const size_t at_sitem = carve(keep, (size_t)ITEM_STRIDE * n_tiles * 64), at_pub = carve(keep, PUB_HDR_WORDS + 4 * (size_t)HB);
    const size_t at_tab = carve(keep, (size_t)7 * n_launch + 4);
    const size_t at_list = carve(keep, std::max<uint32_t>(acc, 1)), at_first = carve(keep, ncls), at_cap = carve(keep, ncls);
    const size_t at_count = carve(keep, 2 * ncls);                         // two sets, used by alternate launches
    const size_t at_jbuf = carve(keep, (size_t)(cut_possible ? n_tiles : 1u) * JOIN_WORDS * 128u);   // JOIN_WORDS 64-bit words per lane of a cut tile
    const size_t at_stats = carve(keep, PS_FIELDS);
    const size_t at_sample = carve(keep, (size_t)(5 * SAMPLE_SLOTS + 1) * n_sample + 64), at_sdist = carve(keep, (size_t)SAMPLE_SLOTS * n_sample + 64),
                 at_stab = carve(keep, (size_t)4 * n_sample + 64);
    MALCS_HIP(malcs_pool_alloc(ctx, (void **)&d.d_rows, keep * sizeof(uint32_t)));
    (void)at_rows;                                                         // the rows come first: d_rows is the allocation
    uint32_t *const base = d.d_rows;
    d.d_pairs = base + at_pairs; d.d_cls = reinterpret_cast<uint8_t *>(base + at_cls); d.d_item = base + at_item;
    d.d_sets = base + at_sets; d.set_words = set_words; d.d_sitem = base + at_sitem; d.d_pub = base + at_pub;
    d.d_piece_off = base + at_tab; d.d_piece_info = d.d_piece_off + n_launch + 1; d.d_piece_join = d.d_piece_info + n_launch;
    d.d_piece_place = d.d_piece_join + n_launch; d.d_lay_off = d.d_piece_place + n_launch; d.d_lay_info = d.d_lay_off + n_launch; d.d_lay_it = d.d_lay_info + n_launch;
    d.d_join_buf = base + at_jbuf; d.d_stats = base + at_stats;
    d.d_retry_list = base + at_list; d.d_retry_first = base + at_first; d.d_retry_cap = base + at_cap; d.d_retry_count = base + at_count;
    d.d_sample = base + at_sample; d.d_sample_dist = reinterpret_cast<int32_t *>(base + at_sdist); d.d_sample_tab = base + at_stab;
    // what starts at zero: the two sets of histogram, cursors and column sums (every plan zeroes the other set for the
    // next), the join buffer (every launch leaves its slots at zero), the statistics
    MALCS_HIP(hipMemsetAsync(d.d_sets, 0, 2 * (size_t)set_words * sizeof(uint32_t), st));
    MALCS_HIP(hipMemsetAsync(d.d_join_buf, 0, (at_stats - at_jbuf + PS_FIELDS) * sizeof(uint32_t), st));
    MALCS_HIP(malcs_h2d(st, d.d_pairs, pairs.data(), count * sizeof(uint32_t)));
    MALCS_HIP(malcs_h2d(st, d.d_cls, retry_class.data(), count));
    if (!classes.empty()) {
        MALCS_HIP(malcs_h2d(st, d.d_retry_first, first.data(), classes.size() * sizeof(uint32_t)));
        MALCS_HIP(malcs_h2d(st, d.d_retry_cap, cap.data(), classes.size() * sizeof(uint32_t)));
    }
    if (n_sample) {
        MALCS_HIP(malcs_h2d(st, d.d_sample, sample.data(), sample.size() * sizeof(uint32_t)));
        const uint32_t subs = SAMPLE_SLOTS * n_sample;
        MALCS_HIP(malcs_h2d(st, d.d_sample + (size_t)(5 * SAMPLE_SLOTS + 1) * n_sample, &subs, sizeof(uint32_t)));
        MALCS_HIP(malcs_h2d(st, d.d_sample_tab, sample_tab.data(), (size_t)4 * n_sample * sizeof(uint32_t)));
    }
    MALCS_HIP(malcs_sync(st));                                             // the sources are locals (copies beyond the staging buffer read them directly)
    d.nlw = nlw; d.T = malcs_diag_threshold(nlw); d.n_tiles = n_tiles; d.n_pairs = count; d.n_launch = n_launch;
    d.max_len = max_len; d.HB = HB; d.can_half = can_half; d.cut_possible = cut_possible; d.n_sample = n_sample;
    d.sample_W = sW; d.sample_L = sL; d.rows_cap = rows_cap; d.parity = 0; d.planned = false; d.plan_parity = 0;
    // one lane per part of a pair (llcs_band32_lane) where the parts are short enough for one lane's time; MALCS_SAMPLE_WAVE=1
    // keeps the full-matrix wave over the host's sub-pairs (round 2's sample; also what longer sequences get)
    d.sample_log_parts = 2;
    while ((LANE_PART_SYMS << d.sample_log_parts) < longest_whole) d.sample_log_parts++;
    d.sample_lanes = n_sample && NP == 2 && (1u << d.sample_log_parts) <= LANE_MAX_PARTS && !(getenv("MALCS_SAMPLE_WAVE") && atoi(getenv("MALCS_SAMPLE_WAVE")) != 0);
    d.sample_inside = n_sample && sW == 1 && sL > 1 && sL <= 64 / (int)SAMPLE_SLOTS;   // (the places of a pair in one wave)                       // the sample as workgroups of the plan's first launch
    return MALCS_OK;
}

int malcs_diag_plan(malcs_batch *b)
{
    diag_plan &d = b->diag;
    if (!d.nlw) return MALCS_OK;
    malcs_ctx *ctx = b->ctx;
    hipStream_t st = ctx->stream;
    const uint32_t count = d.n_pairs, NP = (uint32_t)b->n_planes, HB = d.HB, H2 = 2u * HB;
    plan_args A;
    A.planes = b->d_planes; A.a_off = b->d_a_off; A.a_len = b->d_a_len; A.b_off = b->d_b_off; A.b_len = b->d_b_len;
    A.pairs = d.d_pairs; A.cls = d.d_cls; A.item = d.d_item; A.sets = d.d_sets; A.set_words = d.set_words;
    d.plan_parity ^= 1u;
    A.parity = d.plan_parity;
    A.sitem = d.d_sitem; A.pub = d.d_pub; A.lay_off = d.d_lay_off; A.lay_info = d.d_lay_info; A.lay_it = d.d_lay_it; A.piece_off = d.d_piece_off; A.piece_info = d.d_piece_info;
    A.piece_join = d.d_piece_join; A.piece_place = d.d_piece_place; A.retry_count = d.d_retry_count; A.stats = d.d_stats; A.rows = d.d_rows;
    A.sample = d.d_sample; A.sample_dist = d.d_sample_dist; A.sample_tab = reinterpret_cast<const uint4 *>(d.d_sample_tab); A.sample_log_parts = (uint32_t)d.sample_log_parts;
    A.HB = HB; A.count = count; A.n_tiles = d.n_tiles; A.n_launch = d.n_launch; A.S = (uint32_t)std::max(1, ctx->prop.multiProcessorCount) * 4u;
    A.NP = NP; A.n_sample = d.n_sample; A.n_retry_words = 2u * (uint32_t)std::max<size_t>(d.retry.size(), 1);
    A.min_rows = 32u * (2u * (uint32_t)d.nlw + 2u);
    A.rows_cap = d.rows_cap; A.nlw = d.nlw; A.T = malcs_diag_threshold(d.nlw); A.can_half = d.can_half;
    A.force_half = -1;
    if (const char *env = getenv("MALCS_DIAG_HALF")) if (*env) A.force_half = atoi(env) != 0;
    A.cut_mode = -1;
    if (const char *env = getenv("MALCS_DIAG_CUT")) if (*env) A.cut_mode = atoi(env) != 0;
    if (!d.cut_possible) A.cut_mode = 0;
    A.cut_possible = d.cut_possible;
    // MALCS_DIAG_LDS=0: two-limb pieces fold the match masks from the planes (diag_halves_dna) instead of reading them from LDS
    A.lds_codes = NP == 2 && d.nlw == 3 && !(getenv("MALCS_DIAG_LDS") && atoi(getenv("MALCS_DIAG_LDS")) == 0);
    A.balance = !(getenv("MALCS_DIAG_BALANCE") && atoi(getenv("MALCS_DIAG_BALANCE")) == 0);     // 0: middle cuts only
    A.trim = !(getenv("MALCS_DIAG_TRIM") && atoi(getenv("MALCS_DIAG_TRIM")) == 0);              // 0 keeps the common ends inside the window
    int logL = 0;
    while ((1 << logL) < d.sample_L) logL++;
    A.sample_logL = logL;
    const uint32_t n2 = SAMPLE_SLOTS * d.n_sample, ppw = 64u / (uint32_t)std::max(d.sample_L, 1);
    A.sample_blocks = d.sample_lanes ? (d.n_sample + (256u >> d.sample_log_parts) - 1u) / (256u >> d.sample_log_parts) : d.sample_inside ? ((n2 + ppw - 1u) / ppw + 3u) / 4u : 0u;
    if (d.n_sample && !d.sample_inside && !d.sample_lanes) {
        // sub-pairs beyond 4 096 rows (sequences beyond 8 kb): the sample as a launch of its own, in front of the plan
        uint32_t *sa_off = d.d_sample, *sa_len = sa_off + n2, *sb_off = sa_len + n2, *sb_len = sb_off + n2, *list = sb_len + n2, *cnt = list + n2 + d.n_sample;
        const int rc = malcs_dist_launch_list(st, b->n_planes, d.sample_W, d.sample_L, b->d_planes, sa_off, sa_len, sb_off, sb_len,
                                              list, 0, cnt, n2, (n2 + ppw - 1u) / ppw, d.d_sample_dist);
        if (rc) return rc;
        hipLaunchKernelGGL(k_plan_sample_sums, dim3(1), dim3(256), 0, st, A);
        MALCS_HIP(hipGetLastError());
    }
    const dim3 g_ends(A.sample_blocks + (count + 255u) / 256u), b256(256);
    const size_t hist_lds = (size_t)H2 * sizeof(uint32_t);
    switch (NP) {
    case 2: if (d.sample_lanes) hipLaunchKernelGGL((k_plan_ends<2, 2>), g_ends, b256, hist_lds, st, A);
            else if (d.sample_inside) hipLaunchKernelGGL((k_plan_ends<2, 1>), g_ends, b256, hist_lds, st, A);
            else hipLaunchKernelGGL((k_plan_ends<2, 0>), g_ends, b256, hist_lds, st, A);
            break;
    case 3: hipLaunchKernelGGL((k_plan_ends<3, 0>), g_ends, b256, hist_lds, st, A); break;
    case 4: hipLaunchKernelGGL((k_plan_ends<4, 0>), g_ends, b256, hist_lds, st, A); break;
    }
    MALCS_HIP(hipGetLastError());
    hipLaunchKernelGGL(k_plan_scatter, dim3((count + 1023u) / 1024u + 1u), dim3(1024), 3 * hist_lds, st, A);   // (+ 1: the workgroup that publishes the plan)
    MALCS_HIP(hipGetLastError());
    // the first workgroup deals the pieces to the SIMDs, the others write the rows of two places each
    const dim3 g_layout(1u + (d.n_launch + 1u) / 2u);
    const size_t layout_lds = layout_lds_words(NP, 2u) * sizeof(uint32_t);
    switch (NP) {
    case 2: hipLaunchKernelGGL((k_diag_layout<2, 512>), g_layout, dim3(512), layout_lds, st, A); break;
    case 3: hipLaunchKernelGGL((k_diag_layout<3, 512>), g_layout, dim3(512), layout_lds, st, A); break;
    case 4: hipLaunchKernelGGL((k_diag_layout<4, 512>), g_layout, dim3(512), layout_lds, st, A); break;
    }
    MALCS_HIP(hipGetLastError());
    d.parity = 0;                                                          // the deal has zeroed both sets of retry counters
    d.planned = true;
    return MALCS_OK;
}

// Copies one of the plan's device arrays to the host (tests and dev tools look at the plan with it).  `what`: 0 item records
// in pair order, 1 item records in slot order, 2 lay_off, 3 lay_info, 4 piece_off, 5 piece_info, 6 piece_join,
// 7 piece_place, 8 the published plan.  Returns the number of dwords the array has; copies min(that, cap).
int malcs_diag_dump(malcs_batch *b, int what, uint32_t *out, uint64_t cap, uint64_t *n_out)
{
    diag_plan &d = b->diag;
    *n_out = 0;
    if (!d.nlw || !d.planned) return MALCS_OK;
    const uint32_t *src = nullptr;
    uint64_t n = 0;
    switch (what) {
    case 0: src = d.d_item; n = (uint64_t)ITEM_STRIDE * d.n_pairs; break;
    case 1: src = d.d_sitem; n = (uint64_t)ITEM_STRIDE * d.n_tiles * 64u; break;
    case 2: src = d.d_lay_off; n = d.n_launch; break;
    case 3: src = d.d_lay_info; n = d.n_launch; break;
    case 4: src = d.d_piece_off; n = d.n_launch; break;
    case 5: src = d.d_piece_info; n = d.n_launch; break;
    case 6: src = d.d_piece_join; n = d.n_launch; break;
    case 7: src = d.d_piece_place; n = d.n_launch; break;
    case 8: src = d.d_pub; n = PUB_HDR_WORDS + 4ull * d.HB; break;
    default: malcs_set_error("no such plan array"); return MALCS_ERR_ARG;
    }
    *n_out = n;
    MALCS_HIP(hipStreamSynchronize(b->ctx->stream));
    if (out && cap) MALCS_HIP(hipMemcpy(out, src, std::min(n, cap) * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return MALCS_OK;
}

int malcs_diag_stats(malcs_batch *b, uint32_t *stats)
{
    memset(stats, 0, PS_FIELDS * sizeof(uint32_t));
    diag_plan &d = b->diag;
    if (!d.nlw || !d.planned) return MALCS_OK;
    MALCS_HIP(malcs_d2h(b->ctx->stream, stats, d.d_stats, PS_FIELDS * sizeof(uint32_t)));
    MALCS_HIP(malcs_sync(b->ctx->stream));
    if (stats[PS_ERROR]) { malcs_set_error("window plan failed on the device (code %u)", stats[PS_ERROR]); return MALCS_ERR_INTERNAL; }
    return MALCS_OK;
}

// Launches the window kernel.  The retry counters exist twice: a launch appends through one set and zeroes
// the other one for its successor (stream order makes that safe), which saves a memset per launch.
// `advance` = false re-uses the current set without resetting it.
int malcs_diag_launch(malcs_batch *b, bool advance, hipEvent_t start, hipEvent_t stop)
{
    diag_plan &d = b->diag;
    if (!d.nlw) return MALCS_OK;
    hipStream_t st = b->ctx->stream;
    diag_kernel_t fn = nullptr;
    switch (b->n_planes) {
    case 2: fn = pick_diag<2>(d.nlw); break;
    case 3: fn = pick_diag<3>(d.nlw); break;
    case 4: fn = pick_diag<4>(d.nlw); break;
    }
    if (!fn) { malcs_set_error("no window kernel for %d limbs and %d bit planes", d.nlw, b->n_planes); return MALCS_ERR_INTERNAL; }
    const uint32_t ncls = (uint32_t)std::max<size_t>(d.retry.size(), 1);
    if (ncls > 64) { malcs_set_error("too many retry classes"); return MALCS_ERR_INTERNAL; }
    if (advance) d.parity ^= 1u;
    uint32_t *cur = d.d_retry_count + d.parity * ncls, *next = d.d_retry_count + (d.parity ^ 1u) * ncls;
    unsigned long long *stamps = nullptr;
    const char *stamp_path = getenv("MALCS_DIAG_STAMPS");          // dev: per-wave time stamps of this launch into a file
#ifdef MALCS_DEV
    if (getenv("MALCS_DIAG_NORETRY")) stamps = reinterpret_cast<unsigned long long *>(1);
    else
#endif
    // (the buffer is made by the first launch that asks and kept: a launch right behind a plan must not wait for an allocation)
    static unsigned long long *stamp_buf = nullptr;
    static size_t stamp_cap = 0;
    // MALCS_DIAG_LDS3=0: three-limb pieces do not use their LDS stream (their text rows still hold 2-bit symbols: the
    // plane-folding paths turn them back into planes) -- for the comparison and for the test of that path; 2: the stream
    // stops half way through every piece (the hand-over between the two, which no piece of a plan needs)
    if (const char *e = getenv("MALCS_DIAG_LDS3")) {
        if (atoi(e) == 0) stamps = reinterpret_cast<unsigned long long *>(2);
        else if (atoi(e) == 2) stamps = reinterpret_cast<unsigned long long *>(3);       // half of a piece's blocks through the stream, the rest through the plane paths
    }
    if (stamp_path && *stamp_path) {
        if (stamp_cap < (size_t)d.n_launch * 48) {
            MALCS_HIP(hipStreamSynchronize(st));
            if (stamp_buf) MALCS_HIP(hipFree(stamp_buf));
            stamp_cap = (size_t)d.n_launch * 48;
            MALCS_HIP(hipMalloc((void **)&stamp_buf, stamp_cap));
            MALCS_HIP(hipMemset(stamp_buf, 0, stamp_cap));
        }
        stamps = stamp_buf;
    }
    // The deal balances the blocks of the pieces w, w + S, w + 2 S, ... (S = SIMDs of the device), which is who shares a
    // SIMD when the hardware hands one-wave workgroups out in turn.  It does right behind a launch of the same kind; right
    // behind the layout (workgroups of eight waves) the first turn skips a quarter of every XCD's CUs and doubles up on
    // the others: 92 SIMDs with four pieces, 38 with two, the busiest with 60 blocks instead of 44, the launch 3.5 us
    // longer (tools/diag_in_step.py: per-wave time stamps and where every wave ran).  One workgroup per CU that returns
    // at once, in front of the others in the same launch, takes that first turn: placement is then the deal's in every
    // case (128 such workgroups: not quite; 256 ... 3 126: the same).  MALCS_DIAG_PAD=0 for the comparison.
    uint32_t pad = (uint32_t)std::max(1, b->ctx->prop.multiProcessorCount);
    if (const char *e = getenv("MALCS_DIAG_PAD")) pad = (uint32_t)std::max(0, atoi(e));
    if (start || stop)     // the dispatch's own time stamps (what a profiler reports as the kernel's duration)
        hipExtLaunchKernelGGL(fn, dim3(d.n_launch + pad), dim3(64), 0, st, start, stop, 0, (const uint32_t *)d.d_rows, (const uint32_t *)d.d_piece_off,
                              (const uint32_t *)d.d_piece_info, (const uint32_t *)d.d_piece_join, b->d_dist, d.d_retry_list,
                              (const uint32_t *)d.d_retry_first, (const uint32_t *)d.d_retry_cap, cur, next, advance ? ncls : 0u, d.d_join_buf, stamps, pad);
    else
        hipLaunchKernelGGL(fn, dim3(d.n_launch + pad), dim3(64), 0, st, d.d_rows, d.d_piece_off, d.d_piece_info, d.d_piece_join, b->d_dist,
                           d.d_retry_list, d.d_retry_first, d.d_retry_cap, cur, next, advance ? ncls : 0u, d.d_join_buf, stamps, pad);
    MALCS_HIP(hipGetLastError());
    if (stamps > reinterpret_cast<unsigned long long *>(3)) {
        std::vector<unsigned long long> host((size_t)d.n_launch * 6);
        MALCS_HIP(hipStreamSynchronize(st));
        MALCS_HIP(hipMemcpy(host.data(), stamps, host.size() * 8, hipMemcpyDeviceToHost));
        if (FILE *f = fopen(stamp_path, "wb")) { fwrite(host.data(), 8, host.size(), f); fclose(f); }
    }
    return MALCS_OK;
}
