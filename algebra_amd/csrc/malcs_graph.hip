// malcs_graph.hip -- kernel family 2: all optimal LCS alignments of a pair as maximal match runs.
//
// Stands in for the device-parallel part of LCSgraph.__init__ (reference
// algebra/lcs/lcs_graph.py:59-61): _lcs_nodes (:238-348) emits maximal diagonal match runs bucketed
// by LCS level; _build_graph (:351-453) keeps those that lie on an optimal alignment.  Here the GPU
// produces exactly the runs that lie on some optimal alignment, with their exact levels
// (SURVEY 8(a) P2/P3), and malcs_build.cpp orders and compresses them as the reference does.
//
// Two passes over the DP matrix of a pair (rows = reference positions i, columns = observed
// positions j), both bit-sliced 32 cells per VALU instruction.  Rows are cut into groups of
// G = 64*W rows; a lane owns one group (2W dword limbs in VGPRs).  The lanes that serve a pair form a
// systolic array: in the forward pass group u works on column (step - u), so the carry that leaves a
// group is consumed by the next group one step later and travels as a lane mask in SGPRs -- the same
// machinery as kernel family 1 (malcs_dist.hip); the backward pass mirrors it (groups and columns
// descending).  Short pairs keep all their groups resident and share a wave (L = 2^k lanes per pair);
// long pairs get a wave of their own whose 64 lanes are a window that slides along the band of
// diagonals an optimal alignment can visit (k = col - row in [kmin, kmax], from the exact distance).
//
// forward  (k_lcs_forward)  the bit-vector LCS recurrence.  For every column c and group u inside the
//          band it writes the two tie bits of every cell to HBM -- the 2-bit predecessor mask:
//            T_c[i] = 1  <=>  Lf[i+1][c] == Lf[i][c]     (vertical edge is tight)      = V before column c
//            H_c[i] = 1  <=>  Lf[i][c+1] == Lf[i][c] + 1 (horizontal edge NOT tight)   = carry into bit i
//          (a diagonal edge is tight iff the symbols match).  Groups are stored bit-reversed so that
//          the backward pass can propagate with additions, too.  0.25 B per band cell, 8/16-byte
//          vector stores, the groups of a column contiguous.
//
// backward (k_lcs_reach)  reachability from the sink (m, n) along tight edges, column by column from
//          the right.  With r the reversed row index inside a group:
//            O     = Q & Match'(b_c)                  optimal match cells of column c (Q = R_{c+1} moved one row)
//            seed  = (R_{c+1} & ~H_c) | O             vertices of column c reached from column c+1
//            R_c   = closure of seed through runs of T_c = carries of (seed | T_c) + seed
//          Run boundaries are where O_c and O_{c+1} moved one row differ; only then a wave leaves
//          straight-line code and appends events.  An end-of-run event carries its level
//          Lf[i][c] = base + zero bits of T_c above the row inside the group, where base = Lf at the
//          group's first row is maintained per lane from the stored H bits.  0.25 B per band cell read.
//
// Events (row, col, level, kind) go to a per-pair region (no atomics: the lanes of a pair agree on the
// running count); malcs_build.cpp pairs starts with ends into runs.
#include "malcs_internal.hpp"
#include "malcs_device.hpp"

namespace {

using namespace malcs_dev;

__device__ __forceinline__ uint32_t lane_shr1(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xf, 0xf, false);   // wave_shr:1, lane 0 gets 0
}

template <int NL> struct limb_io;
template <> struct limb_io<2> {
    static __device__ __forceinline__ void store_rev(uint32_t *p, const uint32_t (&x)[2])
    { *reinterpret_cast<uint2 *>(p) = make_uint2(__brev(x[1]), __brev(x[0])); }
    static __device__ __forceinline__ void load(const uint32_t *p, uint32_t (&x)[2])
    { uint2 v = *reinterpret_cast<const uint2 *>(p); x[0] = v.x; x[1] = v.y; }
};
template <int NL> struct limb_io {          // NL = 4, 8, 16, 32: 16-byte vectors
    static __device__ __forceinline__ void store_rev(uint32_t *p, const uint32_t (&x)[NL])
    {
#pragma unroll
        for (int v = 0; v < NL / 4; v++)
            reinterpret_cast<uint4 *>(p)[v] = make_uint4(__brev(x[NL - 1 - 4 * v]), __brev(x[NL - 2 - 4 * v]),
                                                         __brev(x[NL - 3 - 4 * v]), __brev(x[NL - 4 - 4 * v]));
    }
    static __device__ __forceinline__ void load(const uint32_t *p, uint32_t (&x)[NL])
    {
#pragma unroll
        for (int v = 0; v < NL / 4; v++) {
            const uint4 t = reinterpret_cast<const uint4 *>(p)[v];
            x[4 * v] = t.x; x[4 * v + 1] = t.y; x[4 * v + 2] = t.z; x[4 * v + 3] = t.w;
        }
    }
};

// band of row groups stored for column c: groups [ulo, uhi], record slot = u - ulo
__device__ __forceinline__ int band_ulo(int c, int kmax, int logG) { const int v = c - kmax; return v <= 0 ? 0 : (v >> logG); }
__device__ __forceinline__ int band_uhi(int c, int kmin, int logG, int ng) { const int v = (c - kmin) >> logG; return v < ng - 1 ? v : ng - 1; }

// ------------------------------------------------------------------------------------------------
// forward pass
// ------------------------------------------------------------------------------------------------
template <int W, int NP, bool BAND>
__global__ void __launch_bounds__(BAND ? 64 : 256)
k_lcs_forward(const uint32_t *__restrict__ planes,
              const uint32_t *__restrict__ a_off, const uint32_t *__restrict__ a_len,
              const uint32_t *__restrict__ b_off, const uint32_t *__restrict__ b_len,
              const uint32_t *__restrict__ work, uint32_t count, int logL,
              uint32_t *__restrict__ scratch, const uint64_t *__restrict__ sc_off,
              const int32_t *__restrict__ kmin_v, const int32_t *__restrict__ kmax_v, const int32_t *__restrict__ rw_v,
              int32_t *__restrict__ dist_out)
{
    constexpr int NL = 2 * W;
    constexpr int G = 64 * W;
    constexpr int logG = W == 1 ? 6 : (W == 2 ? 7 : (W == 4 ? 8 : (W == 8 ? 9 : 10)));
    const unsigned L = 1u << logL;
    const unsigned lane = threadIdx.x & 63u;
    const unsigned wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const unsigned q = lane & (L - 1u);
    const unsigned slot = wave * (64u >> logL) + (lane >> logL);
    const bool valid = slot < count;
    const unsigned vs = valid ? slot : 0u;
    const uint32_t pid = work[vs];
    const uint32_t m = valid ? a_len[pid] : 0u, n = valid ? b_len[pid] : 0u;
    const uint32_t po = a_off[pid], to = b_off[pid];
    // with a wave per pair these are wave-uniform; say so, the window schedule must stay on the scalar unit
    const int kmin = BAND ? __builtin_amdgcn_readfirstlane(kmin_v[vs]) : kmin_v[vs];
    const int kmax = BAND ? __builtin_amdgcn_readfirstlane(kmax_v[vs]) : kmax_v[vs];
    const int RW = BAND ? __builtin_amdgcn_readfirstlane(rw_v[vs]) : rw_v[vs];
    const int NG = (int)(m / G) + 1;
    uint32_t *const base = scratch + sc_off[vs];
    const uint32_t pblk = (m + 31u) >> 5;
    const int tblk = (int)((n + 31u) >> 5);

    auto sigma_at = [&](int t0) -> int { const int v = t0 - G + 1 - kmax; return (!BAND || v <= 0) ? 0 : (v + G) / (G + 1); };
    auto load_group = [&](uint32_t u, uint32_t (&P)[NL][NP]) {
#pragma unroll
        for (int k = 0; k < NL; k++) {
            const uint32_t g = u * NL + k;
            const bool in = g < pblk;
#pragma unroll
            for (int p = 0; p < NP; p++) P[k][p] = in ? planes[po + g * NP + p] : 0u;
        }
    };
    auto zeros_of = [&](uint32_t u, const uint32_t (&V)[NL]) -> uint32_t {
        uint32_t z = 0;
#pragma unroll
        for (int k = 0; k < NL; k++) {
            const long long vb = (long long)m - (long long)(u * NL + k) * 32ll;
            const uint32_t mask = vb >= 32 ? ~0u : (vb <= 0 ? 0u : ((1u << vb) - 1u));
            z += __popc(~V[k] & mask);
        }
        return z;
    };
    auto load_text = [&](int B, uint32_t u, uint32_t (&tc)[NP], uint32_t (&tp)[NP]) {
        const int blk = B - (int)(u >> 5);
#pragma unroll
        for (int p = 0; p < NP; p++) {
            tc[p] = (blk >= 0 && blk < tblk) ? planes[to + (uint32_t)blk * NP + p] : 0u;
            tp[p] = (blk >= 1 && blk - 1 < tblk) ? planes[to + (uint32_t)(blk - 1) * NP + p] : 0u;
        }
    };

    int sigma = 0;
    uint32_t u = q;
    uint32_t P[NL][NP], V[NL];
    load_group(u, P);
#pragma unroll
    for (int k = 0; k < NL; k++) V[k] = ~0u;
    uint32_t frozen = 0;
    const uint64_t segmask = __builtin_amdgcn_ballot_w64(q != 0u);
    uint64_t cin = zero_mask();

    // the last group (holding vertex row m) stores T_n at step n + NG - 1; all lanes of the wave run until
    // the longest pair of the wave is done
    const int t_end = valid ? (int)n + NG - 1 : -1;
    uint32_t tc[NP], tp[NP];
    load_text(0, u, tc, tp);

    for (int B = 0;; B++) {
        if (__builtin_amdgcn_ballot_w64(32 * B <= t_end) == 0) break;
        const int sigma_next = BAND ? __builtin_amdgcn_readfirstlane(sigma_at(32 * (B + 1))) : 0;
        uint32_t tcn[NP], tpn[NP];
        load_text(B + 1, (uint32_t)sigma_next + q, tcn, tpn);          // prefetch for the next block
        const uint32_t qq = u & 31u;
        uint32_t w[NP];
#pragma unroll
        for (int p = 0; p < NP; p++) w[p] = qq ? __builtin_amdgcn_alignbit(tc[p], tp[p], 32u - qq) : tc[p];
        const int c0 = 32 * B - (int)u;                                  // column of step s is c0 + s
        const long long lo = -(long long)c0;
        const long long hi = (long long)n - c0;
        const uint32_t mlo = lo <= 0 ? ~0u : (lo >= 32 ? 0u : (~0u << lo));
        const uint32_t mhi = hi >= 32 ? ~0u : (hi <= 0 ? 0u : ((1u << hi) - 1u));
        const uint32_t vm = valid ? (mlo & mhi) : 0u;

#pragma unroll 2
        for (int s = 0; s < 32; s++) {
            const int c = c0 + s;
            uint32_t x[NP];
#pragma unroll
            for (int p = 0; p < NP; p++) x[p] = (uint32_t)__builtin_amdgcn_sbfe((int)w[p], s, 1);
            const uint32_t act = (uint32_t)__builtin_amdgcn_sbfe((int)vm, s, 1);
            uint32_t Hc[NL], Vold[NL];
            uint64_t carry = cin;
#pragma unroll
            for (int k = 0; k < NL; k++) {
                uint32_t U, S;
                uint64_t co;
                limb<NP, true>::step(U, S, co, V[k], carry, act, P[k], x);
                Vold[k] = V[k];
                Hc[k] = S ^ V[k] ^ U;                                    // carry into every bit
                V[k] = S | (V[k] ^ U);
                carry = co;
            }
            cin = carry_to_next_lane(carry, segmask);
            // tie bits of (group u, column c) go to HBM if the group lies in the band of that column;
            // column n only has the vertical bits (they close the last column in the backward pass)
            const bool last = valid && c == (int)n;
            if (act || last) {
                const int ulo = band_ulo(c, kmax, logG);
                const int su = (int)u - ulo;
                if (su >= 0 && su < RW && (int)u <= band_uhi(c, kmin, logG, NG)) {
                    // resident windows (all groups, su == u): records in the order of the systolic steps, so that
                    // what the lanes of a pair store in one step is one contiguous stretch (step c + u, group u)
                    const uint64_t row = BAND ? (uint64_t)c : (uint64_t)(c + su);
                    uint32_t *rec = base + (row * (uint64_t)RW + (uint64_t)su) * (2u * NL);
                    limb_io<NL>::store_rev(rec, Vold);
                    if (!last) limb_io<NL>::store_rev(rec + NL, Hc);
                }
            }
        }
        if constexpr (BAND) {
            // slide the window: the top group is frozen, everything moves one lane down
            for (; sigma < sigma_next; sigma++) {
                if (lane == 0) frozen += zeros_of(u, V);
#pragma unroll
                for (int k = 0; k < NL; k++) {
                    V[k] = __shfl_down(V[k], 1);
#pragma unroll
                    for (int p = 0; p < NP; p++) P[k][p] = __shfl_down(P[k][p], 1);
                }
                u += 1;
                if (lane == 63) {
                    load_group(u, P);
#pragma unroll
                    for (int k = 0; k < NL; k++) V[k] = ~0u;
                }
                cin = carry_window_down(cin);
            }
        }
#pragma unroll
        for (int p = 0; p < NP; p++) { tc[p] = tcn[p]; tp[p] = tpn[p]; }
    }

    uint32_t zeros = frozen + zeros_of(u, V);
    for (unsigned s = 1; s < L; s <<= 1) zeros += __shfl_xor(zeros, s);
    if (valid && q == 0) dist_out[slot] = (int32_t)(m + n - 2u * zeros);
}

// ------------------------------------------------------------------------------------------------
// backward pass
// ------------------------------------------------------------------------------------------------
template <int W, int NP, bool BAND>
__global__ void __launch_bounds__(BAND ? 64 : 256)
k_lcs_reach(const uint32_t *__restrict__ planes,
            const uint32_t *__restrict__ a_off, const uint32_t *__restrict__ a_len,
            const uint32_t *__restrict__ b_off, const uint32_t *__restrict__ b_len,
            const uint32_t *__restrict__ work, uint32_t count, int logL,
            const uint32_t *__restrict__ scratch, const uint64_t *__restrict__ sc_off,
            const int32_t *__restrict__ kmin_v, const int32_t *__restrict__ kmax_v, const int32_t *__restrict__ rw_v,
            const int32_t *__restrict__ llcs_v, const uint8_t *__restrict__ skip,
            uint4 *__restrict__ events, const uint64_t *__restrict__ ev_off,
            const uint32_t *__restrict__ ev_cap, uint32_t *__restrict__ ev_cnt)
{
    constexpr int NL = 2 * W;
    constexpr int G = 64 * W;
    constexpr int logG = W == 1 ? 6 : (W == 2 ? 7 : (W == 4 ? 8 : (W == 8 ? 9 : 10)));
    const unsigned L = 1u << logL;
    const unsigned lane = threadIdx.x & 63u;
    const unsigned wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const unsigned qr = lane & (L - 1u);
    const unsigned slot = wave * (64u >> logL) + (lane >> logL);
    const bool valid = slot < count;
    const unsigned vs = valid ? slot : 0u;
    const uint32_t pid = work[vs];
    const bool run = valid && !skip[vs];
    const uint32_t m = a_len[pid];
    const int n = BAND ? __builtin_amdgcn_readfirstlane((int)b_len[pid]) : (int)b_len[pid];
    const uint32_t po = a_off[pid], to = b_off[pid];
    const int kmin = BAND ? __builtin_amdgcn_readfirstlane(kmin_v[vs]) : kmin_v[vs];
    const int kmax = BAND ? __builtin_amdgcn_readfirstlane(kmax_v[vs]) : kmax_v[vs];
    const int RW = BAND ? __builtin_amdgcn_readfirstlane(rw_v[vs]) : rw_v[vs];
    const int llcs = llcs_v[vs];
    const int NG = BAND ? __builtin_amdgcn_readfirstlane((int)(m / G) + 1) : (int)(m / G) + 1;
    const uint32_t *const base_ptr = scratch + sc_off[vs];
    const uint32_t pblk = (m + 31u) >> 5;
    const int tblk = (n + 31) >> 5;
    const int rbit = G - 1 - (int)(m - (uint32_t)(NG - 1) * G);        // reversed position of vertex row m in the last group

    // lowest reverse group index still needed at step t0 (a group stays one step past its first column
    // to flush the run starts it has pending)
    auto varsigma_at = [&](int t0) -> int {
        const long long v = (long long)t0 - n + (long long)G * (NG - 1) + kmin - 1;
        return (!BAND || v <= 0) ? 0 : (int)((v + G) / (G + 1));
    };
    auto load_group_rev = [&](int u, uint32_t (&Pr)[NL][NP]) {
#pragma unroll
        for (int k = 0; k < NL; k++) {
            const long long g = (long long)u * NL + (NL - 1 - k);
            const bool in = u >= 0 && g < (long long)pblk;
#pragma unroll
            for (int p = 0; p < NP; p++) Pr[k][p] = in ? __brev(planes[po + (uint32_t)g * NP + p]) : 0u;
        }
    };
    auto load_text = [&](int Bq, int v, uint32_t (&th)[NP], uint32_t (&tl)[NP]) {
        const int c0 = n - 32 * Bq + v;                                  // column of step s is c0 - s
        const int hb = c0 >> 5;
#pragma unroll
        for (int p = 0; p < NP; p++) {
            th[p] = (hb >= 0 && hb < tblk) ? planes[to + (uint32_t)hb * NP + p] : 0u;
            tl[p] = (hb >= 1 && hb - 1 < tblk) ? planes[to + (uint32_t)(hb - 1) * NP + p] : 0u;
        }
    };

    int varsigma = 0;
    int v = (int)qr;                                                     // reverse group index of this lane
    int u = NG - 1 - v;
    uint32_t Pr[NL][NP];
    load_group_rev(u, Pr);
    uint32_t R[NL], Q[NL], Oprev[NL];
#pragma unroll
    for (int k = 0; k < NL; k++) { R[k] = 0u; Q[k] = 0u; Oprev[k] = 0u; }
    uint32_t h1 = 0, h2 = 0;            // bit of the group's first row in O, one and two steps ago
    int lbase = 0;                      // Lf at the group's first row for the column just processed
    bool was_active = false;
    uint32_t evcount = 0;
    const uint64_t evbase = ev_off[vs];
    const uint32_t evcap = ev_cap[vs];
    const uint64_t segmask = __builtin_amdgcn_ballot_w64(qr != 0u);
    uint64_t cin = zero_mask();

    const int t_end = run ? n + 1 + NG - 1 : -1;
    uint32_t th[NP], tl[NP];
    load_text(0, v, th, tl);

    for (int Bq = 0;; Bq++) {
        if (__builtin_amdgcn_ballot_w64(32 * Bq <= t_end) == 0) break;
        const int varsigma_next = BAND ? __builtin_amdgcn_readfirstlane(varsigma_at(32 * (Bq + 1))) : 0;
        uint32_t thn[NP], tln[NP];
        load_text(Bq + 1, varsigma_next + (int)qr, thn, tln);
        const int c0 = n - 32 * Bq + v;
        const uint32_t off = (uint32_t)c0 & 31u;
        uint32_t wr[NP];
#pragma unroll
        for (int p = 0; p < NP; p++)
            wr[p] = __brev(off == 31u ? th[p] : __builtin_amdgcn_alignbit(th[p], tl[p], off + 1u));   // bit s = column c0 - s

#pragma unroll 2
        for (int s = 0; s < 32; s++) {
            const int c = c0 - s;
            bool active = run && u >= 0 && c >= 0 && c <= n;
            int su = 0;
            if (active) {
                const int ulo = band_ulo(c, kmax, logG);
                su = u - ulo;
                active = su >= 0 && su < RW && u <= band_uhi(c, kmin, logG, NG);
            }
            const bool normal = active && c < n;
            uint32_t T[NL], H[NL];
#pragma unroll
            for (int k = 0; k < NL; k++) { T[k] = 0u; H[k] = ~0u; }
            if (active) {
                const uint64_t row = BAND ? (uint64_t)c : (uint64_t)(c + su);
                const uint32_t *rec = base_ptr + (row * (uint64_t)RW + (uint64_t)su) * (2u * NL);
                limb_io<NL>::load(rec, T);
                if (normal) limb_io<NL>::load(rec + NL, H);
            }
            // Lf at the first row of the group for this column
            const int nbase = (int)lane_shr1((uint32_t)lbase);           // group u+1 finished this column one step ago
            if (active) {
                if (!was_active) {
                    int z = 0;
                    if (u == NG - 1 && c == n) {
#pragma unroll
                        for (int k = 0; k < NL; k++) {
                            const int lowbit = rbit + 1 - 32 * k;          // rows < m are the reversed bits above rbit
                            const uint32_t mask = lowbit <= 0 ? ~0u : (lowbit >= 32 ? 0u : (~0u << lowbit));
                            z += __popc(~T[k] & mask);
                        }
                        lbase = llcs - z;
                    } else {
#pragma unroll
                        for (int k = 0; k < NL; k++) z += __popc(~T[k]);
                        lbase = nbase - z;
                    }
                } else if (normal) {
                    lbase -= (int)(H[NL - 1] >> 31);                     // horizontal step at the group's first row
                }
            }
            was_active = active;

            uint32_t x[NP];
#pragma unroll
            for (int p = 0; p < NP; p++) x[p] = (uint32_t)__builtin_amdgcn_sbfe((int)wr[p], s, 1);
            const uint32_t act = normal ? ~0u : 0u;
            uint32_t O[NL], seed[NL];
#pragma unroll
            for (int k = 0; k < NL; k++) {
                uint32_t o = Q[k] & act;
#pragma unroll
                for (int p = 0; p < NP; p++) o &= ~(Pr[k][p] ^ x[p]);
                O[k] = o;
                seed[k] = (R[k] & ~H[k]) | o;
                if (k == (rbit >> 5) && active && c == n && u == NG - 1) seed[k] |= 1u << (rbit & 31);   // the sink vertex (m, n)
            }

            // run boundaries: O_c against O_{c+1} moved one row (reversed: one bit towards the MSB)
            uint32_t below = lane_shr1(h2);                              // first row of group u+1 in O_{c+1}
            if (qr == 0 || u < 0) below = 0u;                                // row 0 is handled by `row0` in group 0
            uint32_t Osh[NL];
            uint32_t diff = 0;
#pragma unroll
            for (int k = 0; k < NL; k++) {
                const uint32_t lowlimb = k ? (Oprev[k - 1] >> 31) : below;
                Osh[k] = (Oprev[k] << 1) | lowlimb;
                diff |= O[k] ^ Osh[k];
            }
            const bool row0 = (u == 0) && (Oprev[NL - 1] >> 31);         // cell (0, c+1) optimal: a run starts there
            if (__builtin_amdgcn_ballot_w64(diff != 0u || row0)) {
                // ---- rare path: append events ----
                uint32_t ne = row0 ? 1u : 0u;
#pragma unroll
                for (int k = 0; k < NL; k++) ne += __popc(O[k] ^ Osh[k]);
                uint32_t pre = ne;
                for (unsigned d = 1; d < L; d <<= 1) {
                    const uint32_t up = __shfl_up(pre, d, L);
                    if (qr >= d) pre += up;
                }
                const uint32_t total = __shfl(pre, L - 1, L);
                uint32_t idx = evcount + pre - ne;
                const uint32_t row_first = (uint32_t)(u < 0 ? 0 : u) * G;
                uint32_t zabove = 0;
                if (row0) {
                    if (idx < evcap) events[evbase + idx] = make_uint4(0u, (uint32_t)(c + 1), 0u, 1u);
                    idx++;
                }
#pragma unroll
                for (int k = NL - 1; k >= 0; k--) {
                    uint32_t e = O[k] & ~Osh[k];            // run ends in cell (i, c)
                    uint32_t st = Osh[k] & ~O[k];           // run starts in cell (i+1, c+1)
                    const uint32_t nt = ~T[k];
                    while (e) {
                        const uint32_t bit = 31u - (uint32_t)__clz(e);
                        e &= ~(1u << bit);
                        const uint32_t r = (uint32_t)k * 32u + bit;
                        const uint32_t above = bit == 31u ? 0u : __popc(nt >> (bit + 1u));
                        if (idx < evcap)
                            events[evbase + idx] = make_uint4(row_first + (G - 1u - r), (uint32_t)c,
                                                              (uint32_t)(lbase + (int)(zabove + above)), 0u);
                        idx++;
                    }
                    while (st) {
                        const uint32_t bit = 31u - (uint32_t)__clz(st);
                        st &= ~(1u << bit);
                        const uint32_t r = (uint32_t)k * 32u + bit;
                        if (idx < evcap) events[evbase + idx] = make_uint4(row_first + (G - r), (uint32_t)(c + 1), 0u, 1u);
                        idx++;
                    }
                    zabove += __popc(nt);
                }
                evcount += total;
            }
            h2 = h1;
            h1 = O[NL - 1] >> 31;
#pragma unroll
            for (int k = 0; k < NL; k++) Oprev[k] = O[k];

            // closure through the vertical ties: carries of (seed | T) + seed, chained limb to limb and,
            // one step later, group to group
            uint64_t carry = cin;
#pragma unroll
            for (int k = 0; k < NL; k++) {
                const uint32_t xs = seed[k] | T[k];
                uint64_t co;
                const uint32_t S = addc(xs, seed[k], carry, co);
                Q[k] = S ^ xs ^ seed[k];
                R[k] = seed[k] | (T[k] & Q[k]);
                carry = co;
            }
            cin = carry_to_next_lane(carry, segmask);
        }
        if constexpr (BAND) {
            for (; varsigma < varsigma_next; varsigma++) {
#pragma unroll
                for (int k = 0; k < NL; k++) {
                    R[k] = __shfl_down(R[k], 1); Q[k] = __shfl_down(Q[k], 1); Oprev[k] = __shfl_down(Oprev[k], 1);
#pragma unroll
                    for (int p = 0; p < NP; p++) Pr[k][p] = __shfl_down(Pr[k][p], 1);
                }
                h1 = __shfl_down(h1, 1); h2 = __shfl_down(h2, 1);
                lbase = __shfl_down(lbase, 1);
                was_active = __shfl_down((int)was_active, 1) != 0;
                v += 1;
                u -= 1;
                if (lane == 63) {
                    load_group_rev(u, Pr);
#pragma unroll
                    for (int k = 0; k < NL; k++) { R[k] = 0u; Q[k] = 0u; Oprev[k] = 0u; }
                    h1 = 0; h2 = 0; lbase = 0; was_active = false;
                }
                cin = carry_window_down(cin);
            }
        }
#pragma unroll
        for (int p = 0; p < NP; p++) { th[p] = thn[p]; tl[p] = tln[p]; }
    }
    if (valid && qr == 0) ev_cnt[slot] = evcount;
}

// gather the used part of every pair's event region into one contiguous array
__global__ void __launch_bounds__(256)
k_compact_events(const uint4 *__restrict__ events, const uint64_t *__restrict__ ev_off,
                 const uint32_t *__restrict__ ev_n, const uint64_t *__restrict__ out_off,
                 uint32_t count, uint4 *__restrict__ out)
{
    const unsigned wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const unsigned lane = threadIdx.x & 63u;
    if (wave >= count) return;
    const uint32_t n = ev_n[wave];
    const uint4 *src = events + ev_off[wave];
    uint4 *dst = out + out_off[wave];
    for (uint32_t i = lane; i < n; i += 64u) dst[i] = src[i];
}

typedef void (*fwd_kernel_t)(const uint32_t *, const uint32_t *, const uint32_t *, const uint32_t *, const uint32_t *,
                             const uint32_t *, uint32_t, int, uint32_t *, const uint64_t *,
                             const int32_t *, const int32_t *, const int32_t *, int32_t *);
typedef void (*reach_kernel_t)(const uint32_t *, const uint32_t *, const uint32_t *, const uint32_t *, const uint32_t *,
                               const uint32_t *, uint32_t, int, const uint32_t *, const uint64_t *,
                               const int32_t *, const int32_t *, const int32_t *, const int32_t *, const uint8_t *,
                               uint4 *, const uint64_t *, const uint32_t *, uint32_t *);

template <int NP> fwd_kernel_t pick_fwd(int W, bool band)
{
    if (!band) {
        switch (W) {
        case 1: return k_lcs_forward<1, NP, false>;
        case 2: return k_lcs_forward<2, NP, false>;
        case 4: return k_lcs_forward<4, NP, false>;
        default: return nullptr;
        }
    }
    switch (W) {
    case 1: return k_lcs_forward<1, NP, true>;
    case 2: return k_lcs_forward<2, NP, true>;
    case 4: return k_lcs_forward<4, NP, true>;
    case 8: if constexpr (NP <= 3) return k_lcs_forward<8, NP, true>; else return nullptr;
    default: return nullptr;
    }
}
template <int NP> reach_kernel_t pick_reach(int W, bool band)
{
    if (!band) {
        switch (W) {
        case 1: return k_lcs_reach<1, NP, false>;
        case 2: return k_lcs_reach<2, NP, false>;
        case 4: return k_lcs_reach<4, NP, false>;
        default: return nullptr;
        }
    }
    switch (W) {
    case 1: return k_lcs_reach<1, NP, true>;
    case 2: return k_lcs_reach<2, NP, true>;
    case 4: return k_lcs_reach<4, NP, true>;
    case 8: if constexpr (NP <= 3) return k_lcs_reach<8, NP, true>; else return nullptr;
    default: return nullptr;
    }
}

}  // namespace

int malcs_graph_max_band_words(int n_planes) { return n_planes <= 3 ? 8 : 4; }

int malcs_graph_launch_forward(malcs_batch *b, int W, int logL, bool band, const uint32_t *d_work, uint32_t count,
                               uint32_t *d_scratch, const uint64_t *d_sc_off, const int32_t *d_kmin,
                               const int32_t *d_kmax, const int32_t *d_rw, int32_t *d_dist_out)
{
    fwd_kernel_t fn = nullptr;
    switch (b->n_planes) {
    case 2: fn = pick_fwd<2>(W, band); break;
    case 3: fn = pick_fwd<3>(W, band); break;
    case 4: fn = pick_fwd<4>(W, band); break;
    case 8: fn = pick_fwd<8>(W, band); break;
    }
    if (!fn) { malcs_set_error("no forward kernel for W=%d planes=%d band=%d", W, b->n_planes, (int)band); return MALCS_ERR_INTERNAL; }
    const uint32_t ppw = 64u >> logL;
    const uint32_t waves = (count + ppw - 1u) / ppw;
    if (band) hipLaunchKernelGGL(fn, dim3(waves), dim3(64), 0, b->ctx->stream, b->d_planes, b->d_a_off, b->d_a_len, b->d_b_off,
                                 b->d_b_len, d_work, count, logL, d_scratch, d_sc_off, d_kmin, d_kmax, d_rw, d_dist_out);
    else hipLaunchKernelGGL(fn, dim3((waves + 3u) / 4u), dim3(256), 0, b->ctx->stream, b->d_planes, b->d_a_off, b->d_a_len,
                            b->d_b_off, b->d_b_len, d_work, count, logL, d_scratch, d_sc_off, d_kmin, d_kmax, d_rw, d_dist_out);
    MALCS_HIP(hipGetLastError());
    return MALCS_OK;
}

int malcs_graph_launch_reach(malcs_batch *b, int W, int logL, bool band, const uint32_t *d_work, uint32_t count,
                             const uint32_t *d_scratch, const uint64_t *d_sc_off, const int32_t *d_kmin,
                             const int32_t *d_kmax, const int32_t *d_rw, const int32_t *d_llcs, const uint8_t *d_skip,
                             void *d_events, const uint64_t *d_ev_off, const uint32_t *d_ev_cap, uint32_t *d_ev_cnt)
{
    reach_kernel_t fn = nullptr;
    switch (b->n_planes) {
    case 2: fn = pick_reach<2>(W, band); break;
    case 3: fn = pick_reach<3>(W, band); break;
    case 4: fn = pick_reach<4>(W, band); break;
    case 8: fn = pick_reach<8>(W, band); break;
    }
    if (!fn) { malcs_set_error("no reach kernel for W=%d planes=%d band=%d", W, b->n_planes, (int)band); return MALCS_ERR_INTERNAL; }
    const uint32_t ppw = 64u >> logL;
    const uint32_t waves = (count + ppw - 1u) / ppw;
    if (band) hipLaunchKernelGGL(fn, dim3(waves), dim3(64), 0, b->ctx->stream, b->d_planes, b->d_a_off, b->d_a_len, b->d_b_off,
                                 b->d_b_len, d_work, count, logL, d_scratch, d_sc_off, d_kmin, d_kmax, d_rw, d_llcs, d_skip,
                                 (uint4 *)d_events, d_ev_off, d_ev_cap, d_ev_cnt);
    else hipLaunchKernelGGL(fn, dim3((waves + 3u) / 4u), dim3(256), 0, b->ctx->stream, b->d_planes, b->d_a_off, b->d_a_len,
                            b->d_b_off, b->d_b_len, d_work, count, logL, d_scratch, d_sc_off, d_kmin, d_kmax, d_rw, d_llcs,
                            d_skip, (uint4 *)d_events, d_ev_off, d_ev_cap, d_ev_cnt);
    MALCS_HIP(hipGetLastError());
    return MALCS_OK;
}

int malcs_graph_launch_compact(malcs_ctx *ctx, const void *d_events, const uint64_t *d_ev_off, const uint32_t *d_ev_n,
                               const uint64_t *d_out_off, uint32_t count, void *d_out)
{
    if (!count) return MALCS_OK;
    hipLaunchKernelGGL(k_compact_events, dim3((count + 3u) / 4u), dim3(256), 0, ctx->stream, (const uint4 *)d_events,
                       d_ev_off, d_ev_n, d_out_off, count, (uint4 *)d_out);
    MALCS_HIP(hipGetLastError());
    return MALCS_OK;
}
