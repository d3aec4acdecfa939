// malcs_bitpar.hpp -- the full-matrix wave of kernel family 1 (see malcs_dist.hip), shared with the window plan, whose
// sample runs it inside its first kernel.
#pragma once
#include "malcs_device.hpp"

namespace malcs_dev {

// The lanes of a pair work as a systolic array: lane q handles text column (step - q), so the carry
// that leaves lane q-1 in one step is exactly the carry lane q needs in the next step.  It travels
// as a wave-wide lane mask in SGPRs: (carry_out << 1) & segmask, no VALU work and no look-ahead.
#ifndef MALCS_UC
#define MALCS_UC 16
#endif
#ifndef MALCS_MINW
#define MALCS_MINW 8
#endif
// Returns the pair's distance in every one of its lanes (-1 in lanes without a pair).
template <int W, int NP, bool XL>
__device__ __forceinline__ int32_t
llcs_bitpar_wave(const uint32_t *__restrict__ planes,
                 const uint32_t *__restrict__ a_off, const uint32_t *__restrict__ a_len,
                 const uint32_t *__restrict__ b_off, const uint32_t *__restrict__ b_len,
                 const uint32_t *__restrict__ order, uint32_t first, uint32_t count, int logL,
                 int32_t *__restrict__ dist, unsigned wave)
{
    constexpr int NL = 2 * W;
    constexpr int UC = NL <= 8 ? MALCS_UC : (NL <= 16 ? 8 : 4);   // columns unrolled per inner trip
    const unsigned L = 1u << logL;
    const unsigned lane = threadIdx.x & 63u;
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
    const int32_t d = (int32_t)(m + n - 2u * zeros);
    if (valid && q == 0) dist[pid] = d;
    return valid ? d : -1;
}


}  // namespace malcs_dev
