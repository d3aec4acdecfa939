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
//   * the carry between the lanes of a pair travels by DPP (wave_shr:1) as a 0/1 value that is added
//     to the next lane's lowest limb; only if that addition itself overflows (the limb was all ones)
//     does the wave take the ripple path, which finishes the propagation exactly;
//   * text symbols come from the packed arena, NP dwords per 32 columns and lane.
//
// Algorithmic work: 4 dword ops per 32 cells for DNA = 0.125 lane-op per cell.
#include "malcs_internal.hpp"
#include <algorithm>

namespace {

__device__ __forceinline__ uint32_t lane_shr1(uint32_t v)
{
    // lane i receives lane i-1's value, lane 0 receives 0 (wave_shr:1, old = 0)
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xf, 0xf, false);
}

template <int NL, int NP, bool MASKED, bool XL>
__device__ __forceinline__ void column_step(uint32_t (&V)[NL], const uint32_t (&P)[NP][NL],
                                            const uint32_t (&x)[NP], uint32_t act, uint32_t notlast)
{
    uint32_t U[NL], S[NL];
    unsigned c = 0;
#pragma unroll
    for (int k = 0; k < NL; k++) {
        uint32_t u = V[k];
#pragma unroll
        for (int p = 0; p < NP; p++) u &= ~(P[p][k] ^ x[p]);
        if (MASKED) u &= act;
        U[k] = u;
        S[k] = __builtin_addc(V[k], u, c, &c);
    }
    if (XL) {
        uint32_t cv = c ? notlast : 0u;
        uint32_t cin = lane_shr1(cv);
        unsigned co;
        S[0] = __builtin_addc(S[0], cin, 0u, &co);
        if (__builtin_amdgcn_ballot_w64(co != 0)) {
            // Rare: the carry-in overflowed limb 0.  Ripple through the lane's limbs, then keep
            // handing carries to the next lane until none is left (at most L rounds).
            unsigned r = co;
#pragma unroll
            for (int k = 1; k < NL; k++) S[k] = __builtin_addc(S[k], 0u, r, &r);
            uint32_t cv2 = r ? notlast : 0u;
            while (__builtin_amdgcn_ballot_w64(cv2 != 0)) {
                unsigned r2 = lane_shr1(cv2);
#pragma unroll
                for (int k = 0; k < NL; k++) S[k] = __builtin_addc(S[k], 0u, r2, &r2);
                cv2 = r2 ? notlast : 0u;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < NL; k++) V[k] = S[k] | (V[k] ^ U[k]);
}

template <int W, int NP, bool XL>
__global__ void __launch_bounds__(256)
k_llcs_bitpar(const uint32_t *__restrict__ planes,
              const uint32_t *__restrict__ a_off, const uint32_t *__restrict__ a_len,
              const uint32_t *__restrict__ b_off, const uint32_t *__restrict__ b_len,
              const uint32_t *__restrict__ order, uint32_t first, uint32_t count, int logL,
              int32_t *__restrict__ dist)
{
    constexpr int NL = 2 * W;
    constexpr int UC = NL <= 8 ? 32 : (NL <= 16 ? 8 : 4);   // columns unrolled per inner trip
    const unsigned L = 1u << logL;
    const unsigned lane = threadIdx.x & 63u;
    const unsigned wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const unsigned q = lane & (L - 1u);
    const unsigned slot = wave * (64u >> logL) + (lane >> logL);
    const bool valid = slot < count;
    const uint32_t pid = order[first + (valid ? slot : 0u)];
    const uint32_t la = a_len[pid], lb = b_len[pid];
    uint32_t m, n, po, to;
    if (la <= lb) { m = la; n = lb; po = a_off[pid]; to = b_off[pid]; }
    else          { m = lb; n = la; po = b_off[pid]; to = a_off[pid]; }
    if (!valid) { m = 0; n = 0; }

    const uint32_t pblk = (m + 31u) >> 5;
    uint32_t P[NP][NL], V[NL];
#pragma unroll
    for (int k = 0; k < NL; k++) {
        const uint32_t g = q * NL + k;
        const bool in = g < pblk;
#pragma unroll
        for (int p = 0; p < NP; p++) P[p][k] = in ? planes[po + g * NP + p] : 0u;
        V[k] = ~0u;
    }
    const uint32_t notlast = (q != L - 1u) ? 1u : 0u;

    const uint32_t tblk = (n + 31u) >> 5;
    for (uint32_t b = 0;; b++) {
        if (__builtin_amdgcn_ballot_w64(b < tblk) == 0) break;
        uint32_t t[NP];
#pragma unroll
        for (int p = 0; p < NP; p++) t[p] = (b < tblk) ? planes[to + b * NP + p] : 0u;
        const int rem = (int)n - (int)(b * 32u);
        if (__builtin_amdgcn_ballot_w64(rem < 32) == 0) {
#pragma unroll UC
            for (int j = 0; j < 32; j++) {
                uint32_t x[NP];
#pragma unroll
                for (int p = 0; p < NP; p++) x[p] = (uint32_t)__builtin_amdgcn_sbfe((int)t[p], j, 1);
                column_step<NL, NP, false, XL>(V, P, x, 0u, notlast);
            }
        } else {
            // some lane's text ends inside this block: columns past its end must not change V
            const uint32_t vm = rem >= 32 ? ~0u : (rem <= 0 ? 0u : ((1u << rem) - 1u));
#pragma unroll UC
            for (int j = 0; j < 32; j++) {
                uint32_t x[NP];
#pragma unroll
                for (int p = 0; p < NP; p++) x[p] = (uint32_t)__builtin_amdgcn_sbfe((int)t[p], j, 1);
                const uint32_t act = (uint32_t)__builtin_amdgcn_sbfe((int)vm, j, 1);
                column_step<NL, NP, true, XL>(V, P, x, act, notlast);
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
    for (unsigned s = 1; s < L; s <<= 1) zeros += __shfl_xor(zeros, s);
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
