// malcs_dist.hip -- kernel family 1: batched bit-parallel LLCS -> indel-only edit distance.
//
// Stands in for edit_distance(a, b) (reference algebra/lcs/edit_distance.py:21-53), which returns
// m + n - 2 * LLCS(a, b).  The reference walks an O(NP) furthest-point sweep; here every pair runs
// the bit-vector LCS recurrence over the full matrix, 32 cells per VALU instruction:
//
//     U  = V & Match[text symbol]          Match built on the fly from the pattern's bit planes
//     V' = (V + U) | (V ^ U)               (V ^ U == V & ~Match because U is a subset of V)
//     LLCS = number of zero bits of V among the first m positions after the last text symbol
//
// Mapping to the machine (gfx950, wave64):
//   * the shorter sequence is the pattern (rows, bit positions), the longer one the text (columns);
//   * a pair owns L = 2^k adjacent lanes of a wave, each lane owns W 64-bit words (2W dword limbs) of
//     the pattern in VGPRs: bit planes P[plane][limb] and the state V[limb]; 64/L pairs share a wave;
//   * one column costs, per limb, NP v_bitop3_b32 (U), one v_addc_co_u32 (the sum, carry chained
//     through the lane's limbs) and one v_bitop3_b32 (V');
//   * the lanes of a pair form a systolic array: lane q works on text column (step - q), so the carry
//     out of lane q-1 is consumed by lane q one step later.  Carries live in SGPR pairs (lane masks):
//     limb to limb through v_addc_co_u32, lane to lane as (carry_out << 1) & segmask on the scalar
//     unit.  No look-ahead, no second pass, exact for any carry length (carries routinely run through
//     every lane above the DP frontier);
//   * text symbols come from the packed arena, NP dwords per 32 columns and lane.
//
// Algorithmic work: 4 dword ops per 32 cells for DNA = 0.125 lane-op per cell.
#include "malcs_internal.hpp"
#include <algorithm>

namespace {

// One limb of one column: U = V & Match (NP v_bitop3_b32, the match mask never materialises),
// S = V + U + carry (v_addc_co_u32 with the carry in an SGPR pair, so that it can be handed from limb
// to limb, from lane to lane and from step to step without touching VALU registers).
// The two v_bitop3 in front of every v_addc also provide the wait states gfx950 needs between a VALU
// write of an SGPR and a VALU read of it, which the compiler cannot see inside inline asm.
template <int NP, bool MASKED> struct limb;

template <> struct limb<2, false> {
    static __device__ __forceinline__ void step(uint32_t &U, uint32_t &S, uint64_t &cout, uint32_t V, uint64_t cin, uint32_t,
                                                const uint32_t *P, const uint32_t *x)
    {
        asm("v_bitop3_b32 %0, %3, %5, %6 bitop3:0x90\n\tv_bitop3_b32 %0, %0, %7, %8 bitop3:0x90\n\t"
            "v_addc_co_u32_e64 %1, %2, %3, %0, %4"
            : "=&v"(U), "=&v"(S), "=&s"(cout) : "v"(V), "s"(cin), "v"(P[0]), "v"(x[0]), "v"(P[1]), "v"(x[1]));
    }
};
template <> struct limb<2, true> {
    static __device__ __forceinline__ void step(uint32_t &U, uint32_t &S, uint64_t &cout, uint32_t V, uint64_t cin, uint32_t act,
                                                const uint32_t *P, const uint32_t *x)
    {
        asm("v_bitop3_b32 %0, %3, %6, %7 bitop3:0x90\n\tv_bitop3_b32 %0, %0, %8, %9 bitop3:0x90\n\t"
            "v_and_b32 %0, %0, %5\n\tv_addc_co_u32_e64 %1, %2, %3, %0, %4"
            : "=&v"(U), "=&v"(S), "=&s"(cout) : "v"(V), "s"(cin), "v"(act), "v"(P[0]), "v"(x[0]), "v"(P[1]), "v"(x[1]));
    }
};
// generic plane counts: the match mask is folded plane by plane in C (the compiler emits v_bitop3),
// only the carry-chained addition is asm; s_nop 1 supplies the wait states
template <int NP, bool MASKED> struct limb {
    static __device__ __forceinline__ void step(uint32_t &U, uint32_t &S, uint64_t &cout, uint32_t V, uint64_t cin, uint32_t act,
                                                const uint32_t *P, const uint32_t *x)
    {
        uint32_t u = V;
#pragma unroll
        for (int p = 0; p < NP; p++) u &= ~(P[p] ^ x[p]);
        if (MASKED) u &= act;
        U = u;
        asm("s_nop 1\n\tv_addc_co_u32_e64 %0, %1, %2, %3, %4" : "=v"(S), "=&s"(cout) : "v"(V), "v"(u), "s"(cin));
    }
};

// carry out of every lane -> carry into the next lane of the same pair (scalar unit, lane masks).
// Written as asm so that the value provably stays in SGPRs between the v_addc of one step and the next.
__device__ __forceinline__ uint64_t carry_to_next_lane(uint64_t cout, uint64_t segmask)
{
    uint64_t cin;
    asm("s_lshl_b64 %0, %1, 1\n\ts_and_b64 %0, %0, %2" : "=&s"(cin) : "s"(cout), "s"(segmask) : "scc");
    return cin;
}

// One column for one lane: chain the limbs, return the carry out of the lane's top limb.
template <int NL, int NP, bool MASKED>
__device__ __forceinline__ uint64_t column_step(uint32_t (&V)[NL], const uint32_t (&P)[NL][NP],
                                                const uint32_t (&x)[NP], uint32_t act, uint64_t cin)
{
    uint64_t c = cin;
#pragma unroll
    for (int k = 0; k < NL; k++) {
        uint32_t U, S;
        uint64_t co;
        limb<NP, MASKED>::step(U, S, co, V[k], c, act, P[k], x);
        V[k] = S | (V[k] ^ U);
        c = co;
    }
    return c;
}

// The lanes of a pair work as a systolic array: lane q handles text column (step - q), so the carry
// that leaves lane q-1 in one step is exactly the carry lane q needs in the next step.  It travels
// as a wave-wide lane mask in SGPRs: (carry_out << 1) & segmask, no VALU work and no look-ahead.
#ifndef MALCS_UC
#define MALCS_UC 16
#endif
#ifndef MALCS_MINW
#define MALCS_MINW 8
#endif
template <int W, int NP, bool XL>
__global__ void __launch_bounds__(256, MALCS_MINW)
k_llcs_bitpar(const uint32_t *__restrict__ planes,
              const uint32_t *__restrict__ a_off, const uint32_t *__restrict__ a_len,
              const uint32_t *__restrict__ b_off, const uint32_t *__restrict__ b_len,
              const uint32_t *__restrict__ order, uint32_t first, uint32_t count, int logL,
              int32_t *__restrict__ dist)
{
    constexpr int NL = 2 * W;
    constexpr int UC = NL <= 8 ? MALCS_UC : (NL <= 16 ? 8 : 4);   // columns unrolled per inner trip
    const unsigned L = 1u << logL;
    const unsigned lane = threadIdx.x & 63u;
    const unsigned wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const unsigned q = XL ? (lane & (L - 1u)) : 0u;
    const unsigned slot = wave * (64u >> logL) + (lane >> logL);
    const bool valid = slot < count;
    const uint32_t pid = order[first + (valid ? slot : 0u)];
    const uint32_t la = a_len[pid], lb = b_len[pid];
    uint32_t m, n, po, to;
    if (la <= lb) { m = la; n = lb; po = a_off[pid]; to = b_off[pid]; }
    else          { m = lb; n = la; po = b_off[pid]; to = a_off[pid]; }
    if (!valid) { m = 0; n = 0; }

    const uint32_t pblk = (m + 31u) >> 5;
    uint32_t P[NL][NP], V[NL];
#pragma unroll
    for (int k = 0; k < NL; k++) {
        const uint32_t g = q * NL + k;
        const bool in = g < pblk;
#pragma unroll
        for (int p = 0; p < NP; p++) P[k][p] = in ? planes[po + g * NP + p] : 0u;
        V[k] = ~0u;
    }
    const uint64_t segmask = XL ? __builtin_amdgcn_ballot_w64(q != 0u) : 0ull;
    uint64_t zero64;                               // an SGPR pair holding 0 (the asm needs a register, not a literal)
    asm("s_mov_b64 %0, 0" : "=s"(zero64));
    uint64_t cin = zero64;

    const int tblk = (int)((n + 31u) >> 5);
    const int qb = (int)(q >> 5);                  // this lane lags the pair's first lane by q columns
    const uint32_t qq = q & 31u;
    uint32_t tprev[NP], tnext[NP];                 // text block of the previous / of the coming block of steps
#pragma unroll
    for (int p = 0; p < NP; p++) {
        tprev[p] = 0u;
        tnext[p] = (qb == 0 && tblk > 0) ? planes[to + p] : 0u;
    }

    for (int B = 0;; B++) {
        const int lo = (int)q - 32 * B;            // first and one-past-last valid step of this lane in the block
        const int hi = (int)(n + q) - 32 * B;
        if (__builtin_amdgcn_ballot_w64(hi > 0) == 0) break;
        const int nblk = B + 1 - qb;               // text block whose columns start inside the NEXT block of steps
        uint32_t w[NP];
#pragma unroll
        for (int p = 0; p < NP; p++) {
            const uint32_t tcur = tnext[p];
            // prefetch one block ahead so that the load latency hides behind 32 columns of VALU work
            tnext[p] = (nblk >= 0 && nblk < tblk) ? planes[to + (uint32_t)nblk * NP + p] : 0u;
            // bit s of w = text bit of column 32*B + s - q
            w[p] = qq ? __builtin_amdgcn_alignbit(tcur, tprev[p], 32u - qq) : tcur;
            tprev[p] = tcur;
        }
        const uint32_t mlo = lo <= 0 ? ~0u : (lo >= 32 ? 0u : (~0u << lo));
        const uint32_t mhi = hi >= 32 ? ~0u : (hi <= 0 ? 0u : ((1u << hi) - 1u));
        const uint32_t vm = mlo & mhi;
        if (__builtin_amdgcn_ballot_w64(vm != ~0u) == 0) {
#pragma unroll UC
            for (int s = 0; s < 32; s++) {
                uint32_t x[NP];
#pragma unroll
                for (int p = 0; p < NP; p++) x[p] = (uint32_t)__builtin_amdgcn_sbfe((int)w[p], s, 1);
                const uint64_t co = column_step<NL, NP, false>(V, P, x, 0u, cin);
                cin = XL ? carry_to_next_lane(co, segmask) : zero64;
            }
        } else {
            // first / last blocks: steps outside [0, n) of a lane's own columns must not change its V
#pragma unroll UC
            for (int s = 0; s < 32; s++) {
                uint32_t x[NP];
#pragma unroll
                for (int p = 0; p < NP; p++) x[p] = (uint32_t)__builtin_amdgcn_sbfe((int)w[p], s, 1);
                const uint32_t act = (uint32_t)__builtin_amdgcn_sbfe((int)vm, s, 1);
                const uint64_t co = column_step<NL, NP, true>(V, P, x, act, cin);
                cin = XL ? carry_to_next_lane(co, segmask) : zero64;
            }
        }
    }

    // LLCS = zero bits of V among the pattern's m positions; reduce over the pair's lanes
    uint32_t zeros = 0;
#pragma unroll
    for (int k = 0; k < NL; k++) {
        const int vb = (int)m - (int)((q * NL + k) * 32u);
        const uint32_t mask = vb >= 32 ? ~0u : (vb <= 0 ? 0u : ((1u << vb) - 1u));
        zeros += __popc(~V[k] & mask);
    }
    if (XL) for (unsigned s = 1; s < L; s <<= 1) zeros += __shfl_xor(zeros, s);
    if (valid && q == 0) dist[pid] = (int32_t)(m + n - 2u * zeros);
}

typedef void (*dist_kernel_t)(const uint32_t *, const uint32_t *, const uint32_t *, const uint32_t *,
                              const uint32_t *, const uint32_t *, uint32_t, uint32_t, int, int32_t *);

template <int NP>
dist_kernel_t pick_kernel(int W, bool xl)
{
    switch (W) {
    case 1: return xl ? k_llcs_bitpar<1, NP, true> : k_llcs_bitpar<1, NP, false>;
    case 2: return xl ? k_llcs_bitpar<2, NP, true> : k_llcs_bitpar<2, NP, false>;
    case 4: return xl ? k_llcs_bitpar<4, NP, true> : k_llcs_bitpar<4, NP, false>;
    case 8: if constexpr (NP <= 4) return k_llcs_bitpar<8, NP, true>; else return nullptr;
    case 16: if constexpr (NP <= 3) return k_llcs_bitpar<16, NP, true>; else return nullptr;
    default: return nullptr;
    }
}

int max_words_per_lane(int n_planes) { return n_planes <= 3 ? 16 : (n_planes == 4 ? 8 : 4); }

uint32_t pow2ceil(uint32_t x) { uint32_t p = 1; while (p < x) p <<= 1; return p; }

}  // namespace

// Group the pairs of a batch into launch classes (W words per lane, L lanes per pair).
int malcs_dist_plan(malcs_batch *b)
{
    const uint64_t n = b->n_pairs;
    const int wmax = max_words_per_lane(b->n_planes);
    std::vector<uint8_t> cls(n);
    // class id = index into table of (W, logL)
    struct wl { int W, logL; };
    std::vector<wl> table;
    auto class_of = [&](int W, int logL) -> int {
        for (size_t i = 0; i < table.size(); i++) if (table[i].W == W && table[i].logL == logL) return (int)i;
        table.push_back({W, logL});
        return (int)table.size() - 1;
    };
    for (uint64_t i = 0; i < n; i++) {
        uint32_t m = std::min(b->a_len[i], b->b_len[i]);
        uint32_t words = std::max<uint32_t>(1u, (m + 63u) / 64u);
        int W, logL = 0;
        if (words <= 4 && (int)words <= wmax) {
            W = (int)pow2ceil(words);
        } else {
            W = 4;
            while ((uint32_t)W * 64u < words && W < wmax) W *= 2;
            if ((uint32_t)W * 64u < words) {
                malcs_set_error("pair %llu: shorter sequence has %u symbols; the distance kernels hold at most %u "
                                "symbols of it with %d bit planes", (unsigned long long)i, m, (uint32_t)wmax * 64u * 64u,
                                b->n_planes);
                return MALCS_ERR_TOO_LONG;
            }
            uint32_t lanes = pow2ceil((words + (uint32_t)W - 1u) / (uint32_t)W);
            while ((1u << logL) < lanes) logL++;
        }
        cls[i] = (uint8_t)class_of(W, logL);
    }
    std::vector<uint32_t> cnt(table.size() + 1, 0);
    for (uint64_t i = 0; i < n; i++) cnt[cls[i] + 1]++;
    for (size_t c = 0; c < table.size(); c++) cnt[c + 1] += cnt[c];
    std::vector<uint32_t> order(n), pos(cnt.begin(), cnt.end() - 1);
    for (uint64_t i = 0; i < n; i++) order[pos[cls[i]]++] = (uint32_t)i;
    b->dist_buckets.clear();
    for (size_t c = 0; c < table.size(); c++) {
        dist_bucket k;
        k.W = table[c].W; k.L = 1 << table[c].logL; k.first = cnt[c]; k.count = cnt[c + 1] - cnt[c];
        if (k.count) b->dist_buckets.push_back(k);
    }
    if (n) {
        MALCS_HIP(hipMalloc(&b->d_order, n * sizeof(uint32_t)));
        MALCS_HIP(hipMemcpyAsync(b->d_order, order.data(), n * sizeof(uint32_t), hipMemcpyHostToDevice, b->ctx->stream));
        MALCS_HIP(hipStreamSynchronize(b->ctx->stream));   // `order` is a local
    }
    return MALCS_OK;
}

int malcs_dist_launch(malcs_batch *b)
{
    for (const dist_bucket &k : b->dist_buckets) {
        int logL = 0;
        while ((1 << logL) < k.L) logL++;
        const bool xl = k.L > 1;
        dist_kernel_t fn = nullptr;
        switch (b->n_planes) {
        case 2: fn = pick_kernel<2>(k.W, xl); break;
        case 3: fn = pick_kernel<3>(k.W, xl); break;
        case 4: fn = pick_kernel<4>(k.W, xl); break;
        case 8: fn = pick_kernel<8>(k.W, xl); break;
        }
        if (!fn) { malcs_set_error("no distance kernel for W=%d planes=%d", k.W, b->n_planes); return MALCS_ERR_INTERNAL; }
        const uint32_t ppw = 64u / (uint32_t)k.L;
        const uint32_t waves = (k.count + ppw - 1u) / ppw;
        const uint32_t blocks = (waves + 3u) / 4u;
        hipLaunchKernelGGL(fn, dim3(blocks), dim3(256), 0, b->ctx->stream, b->d_planes, b->d_a_off, b->d_a_len,
                           b->d_b_off, b->d_b_len, b->d_order, k.first, k.count, logL, b->d_dist);
        MALCS_HIP(hipGetLastError());
    }
    return MALCS_OK;
}
