// malcs_graph.hip -- kernel family 2: all optimal LCS alignments of a pair as maximal match runs.
//
// Stands in for the device-parallel part of LCSgraph.__init__ (reference
// algebra/lcs/lcs_graph.py:59-61): _lcs_nodes (:238-348) emits maximal diagonal match runs bucketed
// by LCS level; _build_graph (:351-453) keeps those that lie on an optimal alignment.  Here the GPU
// produces exactly the runs that lie on some optimal alignment, with their exact levels
// (SURVEY 8(a) P2/P3), and malcs_build.cpp orders and compresses them as the reference does.
//
// Two passes over the DP matrix of a pair (rows = reference positions i, columns = observed
// positions j), both bit-sliced 32 cells per VALU instruction, rows spread over the L lanes of the
// pair (W 64-bit words per lane), several pairs per wave -- the same mapping as family 1:
//
// forward  (k_lcs_forward)  the bit-vector LCS recurrence, column by column.  For every column it
//          writes the two tie bits of every cell to HBM -- the 2-bit predecessor mask:
//            T_j[i] = 1  <=>  Lf[i+1][j] == Lf[i][j]   (vertical edge (i,j)->(i+1,j) is tight) = V before column j
//            H_j[i] = 1  <=>  Lf[i][j+1] == Lf[i][j]+1 (horizontal edge (i,j)->(i,j+1) is NOT tight) = carry into bit i
//          (a diagonal edge is tight iff the symbols match).  Rows are stored bit-reversed and
//          limb-mirrored so that the backward pass can propagate with additions, too.
//          0.25 B per cell written, coalesced (each lane stores its limbs as 8/16-byte vectors,
//          the lanes of a pair cover a contiguous row).
//
// backward (k_lcs_reach)  reachability from the sink (m, n) along tight edges, column by column from
//          the right.  With r = NB-1-i (reversed row index):
//            O     = Q & Match'(b_j)                  optimal match cells of column j  (Q = R_{j+1} shifted by one row)
//            seed  = (R_{j+1} & ~H_j) | O             vertices of column j reached from column j+1
//            R_j   = closure of seed through runs of T_j = carries of (seed | T_j) + seed
//          The closure needs carries that run through whole lanes, so lanes exchange generate /
//          propagate masks by ballot and resolve them with one scalar carry-lookahead addition.
//          Run boundaries are where O_j and O_{j+1} shifted by one row differ; a wave only leaves the
//          straight-line code when some lane sees a boundary.  An end-of-run event carries the level
//          Lf[i][j] = number of zero bits of T_j below row i (popcounts + a segmented lane scan).
//          0.25 B per cell read.
//
// Events (row, col, level, kind) are appended to a per-pair region (no atomics: the lanes of a pair
// agree on the running count); malcs_build.cpp pairs starts with ends into runs.
#include "malcs_internal.hpp"

namespace {

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
template <> struct limb_io<4> {
    static __device__ __forceinline__ void store_rev(uint32_t *p, const uint32_t (&x)[4])
    { *reinterpret_cast<uint4 *>(p) = make_uint4(__brev(x[3]), __brev(x[2]), __brev(x[1]), __brev(x[0])); }
    static __device__ __forceinline__ void load(const uint32_t *p, uint32_t (&x)[4])
    { uint4 v = *reinterpret_cast<const uint4 *>(p); x[0] = v.x; x[1] = v.y; x[2] = v.z; x[3] = v.w; }
};
template <> struct limb_io<8> {
    static __device__ __forceinline__ void store_rev(uint32_t *p, const uint32_t (&x)[8])
    {
        reinterpret_cast<uint4 *>(p)[0] = make_uint4(__brev(x[7]), __brev(x[6]), __brev(x[5]), __brev(x[4]));
        reinterpret_cast<uint4 *>(p)[1] = make_uint4(__brev(x[3]), __brev(x[2]), __brev(x[1]), __brev(x[0]));
    }
    static __device__ __forceinline__ void load(const uint32_t *p, uint32_t (&x)[8])
    {
        uint4 v = reinterpret_cast<const uint4 *>(p)[0], w = reinterpret_cast<const uint4 *>(p)[1];
        x[0] = v.x; x[1] = v.y; x[2] = v.z; x[3] = v.w; x[4] = w.x; x[5] = w.y; x[6] = w.z; x[7] = w.w;
    }
};

// ------------------------------------------------------------------------------------------------
// forward pass
// ------------------------------------------------------------------------------------------------
template <int W, int NP>
__global__ void __launch_bounds__(256)
k_lcs_forward(const uint32_t *__restrict__ planes,
              const uint32_t *__restrict__ a_off, const uint32_t *__restrict__ a_len,
              const uint32_t *__restrict__ b_off, const uint32_t *__restrict__ b_len,
              const uint32_t *__restrict__ work, uint32_t count, int logL,
              uint32_t *__restrict__ scratch, const uint64_t *__restrict__ sc_off,
              int32_t *__restrict__ dist)
{
    constexpr int NL = 2 * W;
    const unsigned L = 1u << logL;
    const unsigned lane = threadIdx.x & 63u;
    const unsigned wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const unsigned q = lane & (L - 1u);
    const unsigned slot = wave * (64u >> logL) + (lane >> logL);
    const bool valid = slot < count;
    const uint32_t pid = work[valid ? slot : 0u];
    const uint32_t m = valid ? a_len[pid] : 0u, n = valid ? b_len[pid] : 0u;
    const uint32_t po = a_off[pid], to = b_off[pid];
    const uint32_t NLtot = (uint32_t)NL << logL;
    const uint64_t rowstride = 2ull * NLtot;
    uint32_t *const base = scratch + sc_off[valid ? slot : 0u] + (NLtot - (q + 1u) * NL);

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
#pragma unroll 4
        for (int j = 0; j < 32; j++) {
            const uint32_t col = b * 32u + (uint32_t)j;
            const bool active = col < n;
            const uint32_t act = active ? ~0u : 0u;
            uint32_t x[NP];
#pragma unroll
            for (int p = 0; p < NP; p++) x[p] = (uint32_t)__builtin_amdgcn_sbfe((int)t[p], j, 1);
            uint32_t U[NL], S[NL];
            unsigned c = 0;
#pragma unroll
            for (int k = 0; k < NL; k++) {
                uint32_t u = V[k] & act;
#pragma unroll
                for (int p = 0; p < NP; p++) u &= ~(P[p][k] ^ x[p]);
                U[k] = u;
                S[k] = __builtin_addc(V[k], u, c, &c);
            }
            if (L > 1) {
                uint32_t cv = c ? notlast : 0u;
                uint32_t cin = lane_shr1(cv);
                unsigned co;
                S[0] = __builtin_addc(S[0], cin, 0u, &co);
                if (__builtin_amdgcn_ballot_w64(co != 0)) {
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
            uint32_t Hc[NL];
#pragma unroll
            for (int k = 0; k < NL; k++) Hc[k] = S[k] ^ V[k] ^ U[k];       // carry into every bit
            if (active) {
                uint32_t *row = base + (uint64_t)col * rowstride;
                limb_io<NL>::store_rev(row, V);
                limb_io<NL>::store_rev(row + NLtot, Hc);
            }
#pragma unroll
            for (int k = 0; k < NL; k++) V[k] = S[k] | (V[k] ^ U[k]);
        }
    }
    if (valid) limb_io<NL>::store_rev(base + (uint64_t)n * rowstride, V);   // T_n: closes the last column

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

// ------------------------------------------------------------------------------------------------
// backward pass
// ------------------------------------------------------------------------------------------------

// R = closure of `seed` through runs of ones of T towards higher (reversed) bit positions,
// Q = R shifted up by one position.  Implemented as the carries of (seed | T) + seed.
template <int NL>
__device__ __forceinline__ void closure(const uint32_t (&seed)[NL], const uint32_t (&T)[NL],
                                        uint32_t (&R)[NL], uint32_t (&Q)[NL],
                                        unsigned lane, uint64_t lastmask, bool multi)
{
    uint32_t x[NL], S[NL];
    unsigned c = 0;
    uint32_t all = ~0u;
#pragma unroll
    for (int k = 0; k < NL; k++) {
        x[k] = seed[k] | T[k];
        S[k] = __builtin_addc(x[k], seed[k], c, &c);
        all &= S[k];
    }
    if (multi) {
        const uint64_t G = __builtin_amdgcn_ballot_w64(c != 0) & ~lastmask;
        const uint64_t Pm = __builtin_amdgcn_ballot_w64(all == ~0u) & ~lastmask;
        const uint64_t cinmask = ((G | Pm) + G) ^ Pm;           // carry into every lane
        unsigned ci = (unsigned)((cinmask >> lane) & 1ull);
#pragma unroll
        for (int k = 0; k < NL; k++) S[k] = __builtin_addc(S[k], 0u, ci, &ci);
    }
#pragma unroll
    for (int k = 0; k < NL; k++) {
        Q[k] = S[k] ^ x[k] ^ seed[k];
        R[k] = seed[k] | (T[k] & Q[k]);
    }
}

template <int W, int NP>
__global__ void __launch_bounds__(256)
k_lcs_reach(const uint32_t *__restrict__ planes,
            const uint32_t *__restrict__ a_off, const uint32_t *__restrict__ a_len,
            const uint32_t *__restrict__ b_off, const uint32_t *__restrict__ b_len,
            const uint32_t *__restrict__ work, uint32_t count, int logL,
            const uint32_t *__restrict__ scratch, const uint64_t *__restrict__ sc_off,
            const uint8_t *__restrict__ skip,
            uint4 *__restrict__ events, const uint64_t *__restrict__ ev_off,
            const uint32_t *__restrict__ ev_cap, uint32_t *__restrict__ ev_cnt)
{
    constexpr int NL = 2 * W;
    const unsigned L = 1u << logL;
    const unsigned lane = threadIdx.x & 63u;
    const unsigned wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const unsigned q = lane & (L - 1u);
    const unsigned slot = wave * (64u >> logL) + (lane >> logL);
    const bool valid = slot < count;
    const uint32_t pid = work[valid ? slot : 0u];
    const bool run = valid && !skip[valid ? slot : 0u];
    const uint32_t m = a_len[pid];
    const int n = run ? (int)b_len[pid] : -2;                 // -2: nothing to do for this lane group
    const uint32_t po = a_off[pid], to = b_off[pid];
    const uint32_t NLtot = (uint32_t)NL << logL;
    const uint32_t NB = 32u * NLtot;
    const uint64_t rowstride = 2ull * NLtot;
    const uint32_t *const base = scratch + sc_off[valid ? slot : 0u] + q * NL;
    const bool multi = L > 1;
    const uint64_t lastmask = __builtin_amdgcn_ballot_w64(q == L - 1u);

    // reversed pattern planes: reversed limb g' holds rows of original limb NLtot-1-g', bit-reversed
    const uint32_t pblk = (m + 31u) >> 5;
    uint32_t Pr[NP][NL];
#pragma unroll
    for (int k = 0; k < NL; k++) {
        const uint32_t g = NLtot - 1u - (q * NL + k);
        const bool in = g < pblk;
#pragma unroll
        for (int p = 0; p < NP; p++) Pr[p][k] = in ? __brev(planes[po + g * NP + p]) : 0u;
    }

    uint32_t R[NL], Q[NL], Oprev[NL], T[NL], H[NL];
    {
        // column n: the sink vertex (m, n), closed through T_n
        const uint32_t rm = NB - 1u - m;
        uint32_t seed[NL];
#pragma unroll
        for (int k = 0; k < NL; k++) {
            seed[k] = (run && (q * NL + k) == (rm >> 5)) ? (1u << (rm & 31u)) : 0u;
            Oprev[k] = 0u;
            T[k] = 0u;
        }
        if (run) limb_io<NL>::load(base + (uint64_t)n * rowstride, T);
        closure<NL>(seed, T, R, Q, lane, lastmask, multi);
    }

    uint32_t evcount = 0;
    const uint64_t evbase = ev_off[valid ? slot : 0u];
    const uint32_t evcap = ev_cap[valid ? slot : 0u];
    uint32_t tcur[NP];
#pragma unroll
    for (int p = 0; p < NP; p++) tcur[p] = 0u;

    for (int t = 0;; t++) {
        const int j = n - 1 - t;                                // this lane group's column; -1 = flush
        if (__builtin_amdgcn_ballot_w64(j >= -1) == 0) break;
        const bool normal = j >= 0;
        uint32_t O[NL];
        if (normal) {
            const uint32_t *row = base + (uint64_t)j * rowstride;
            limb_io<NL>::load(row, T);
            limb_io<NL>::load(row + NLtot, H);
            if (t == 0 || (j & 31) == 31) {
#pragma unroll
                for (int p = 0; p < NP; p++) tcur[p] = planes[to + (uint32_t)(j >> 5) * NP + p];
            }
        } else {
#pragma unroll
            for (int k = 0; k < NL; k++) { T[k] = 0u; H[k] = 0u; }
        }
        uint32_t x[NP];
#pragma unroll
        for (int p = 0; p < NP; p++) x[p] = (uint32_t)__builtin_amdgcn_sbfe((int)tcur[p], (unsigned)(j & 31), 1u);
        const uint32_t act = normal ? ~0u : 0u;
        uint32_t seed[NL];
#pragma unroll
        for (int k = 0; k < NL; k++) {
            uint32_t o = Q[k] & act;
#pragma unroll
            for (int p = 0; p < NP; p++) o &= ~(Pr[p][k] ^ x[p]);
            O[k] = o;
            seed[k] = (R[k] & ~H[k]) | o;
        }
        // run boundaries: O_j against O_{j+1} moved one row up (reversed: one bit towards the MSB)
        uint32_t below = lane_shr1(Oprev[NL - 1]);
        if (q == 0) below = 0u;
        uint32_t Osh[NL];
        uint32_t diff = 0;
#pragma unroll
        for (int k = 0; k < NL; k++) {
            const uint32_t lo = k ? Oprev[k - 1] : below;
            Osh[k] = (Oprev[k] << 1) | (lo >> 31);
            diff |= O[k] ^ Osh[k];
        }
        const bool row0 = (q == L - 1u) && (Oprev[NL - 1] >> 31);     // cell (0, j+1) optimal: a run starts there
        if (__builtin_amdgcn_ballot_w64(diff != 0u || row0)) {
            // ---- rare path: emit events ----
            uint32_t ne = row0 ? 1u : 0u, lz = 0;
#pragma unroll
            for (int k = 0; k < NL; k++) { ne += __popc(O[k] ^ Osh[k]); lz += __popc(~T[k]); }
            // segmented scans over the lanes of the pair
            uint32_t pre = ne, sfx = lz;
            for (unsigned d = 1; d < L; d <<= 1) {
                const uint32_t up = __shfl_up(pre, d, L);
                const uint32_t dn = __shfl_down(sfx, d, L);
                if (q >= d) pre += up;
                if (q + d < L) sfx += dn;
            }
            const uint32_t total = __shfl(pre, L - 1, L);
            uint32_t idx = evcount + pre - ne;          // exclusive prefix
            uint32_t zabove = sfx - lz;                 // zero bits of T in higher lanes (= lower rows)
            if (row0) {
                if (idx < evcap) events[evbase + idx] = make_uint4(0u, (uint32_t)(j + 1), 0u, 1u);
                idx++;
            }
#pragma unroll
            for (int k = NL - 1; k >= 0; k--) {
                uint32_t e = O[k] & ~Osh[k];            // run ends in cell (i, j)
                uint32_t s = Osh[k] & ~O[k];            // run starts in cell (i+1, j+1)
                const uint32_t nt = ~T[k];
                while (e) {
                    const uint32_t bit = 31u - (uint32_t)__clz(e);
                    e &= ~(1u << bit);
                    const uint32_t r = (q * NL + k) * 32u + bit;
                    const uint32_t above = bit == 31u ? 0u : __popc(nt >> (bit + 1u));
                    if (idx < evcap) events[evbase + idx] = make_uint4(NB - 1u - r, (uint32_t)j, zabove + above, 0u);
                    idx++;
                }
                while (s) {
                    const uint32_t bit = 31u - (uint32_t)__clz(s);
                    s &= ~(1u << bit);
                    const uint32_t r = (q * NL + k) * 32u + bit;
                    if (idx < evcap) events[evbase + idx] = make_uint4(NB - r, (uint32_t)(j + 1), 0u, 1u);
                    idx++;
                }
                zabove += __popc(nt);
            }
            evcount += total;
        }
#pragma unroll
        for (int k = 0; k < NL; k++) Oprev[k] = O[k];
        uint32_t Rn[NL], Qn[NL];
        closure<NL>(seed, T, Rn, Qn, lane, lastmask, multi);
#pragma unroll
        for (int k = 0; k < NL; k++) { R[k] = Rn[k]; Q[k] = Qn[k]; }
    }
    if (valid && q == 0) ev_cnt[slot] = evcount;
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
                             const uint32_t *, uint32_t, int, uint32_t *, const uint64_t *, int32_t *);
typedef void (*reach_kernel_t)(const uint32_t *, const uint32_t *, const uint32_t *, const uint32_t *, const uint32_t *,
                               const uint32_t *, uint32_t, int, const uint32_t *, const uint64_t *, const uint8_t *,
                               uint4 *, const uint64_t *, const uint32_t *, uint32_t *);

template <int NP> fwd_kernel_t pick_fwd(int W)
{
    switch (W) {
    case 1: return k_lcs_forward<1, NP>;
    case 2: return k_lcs_forward<2, NP>;
    case 4: return k_lcs_forward<4, NP>;
    default: return nullptr;
    }
}
template <int NP> reach_kernel_t pick_reach(int W)
{
    switch (W) {
    case 1: return k_lcs_reach<1, NP>;
    case 2: return k_lcs_reach<2, NP>;
    case 4: return k_lcs_reach<4, NP>;
    default: return nullptr;
    }
}

}  // namespace

int malcs_graph_launch_forward(malcs_batch *b, int W, int logL, const uint32_t *d_work, uint32_t count,
                               uint32_t *d_scratch, const uint64_t *d_sc_off, int32_t *d_dist)
{
    fwd_kernel_t fn = nullptr;
    switch (b->n_planes) {
    case 2: fn = pick_fwd<2>(W); break;
    case 3: fn = pick_fwd<3>(W); break;
    case 4: fn = pick_fwd<4>(W); break;
    case 8: fn = pick_fwd<8>(W); break;
    }
    if (!fn) { malcs_set_error("no forward kernel for W=%d planes=%d", W, b->n_planes); return MALCS_ERR_INTERNAL; }
    const uint32_t ppw = 64u >> logL;
    const uint32_t waves = (count + ppw - 1u) / ppw;
    hipLaunchKernelGGL(fn, dim3((waves + 3u) / 4u), dim3(256), 0, b->ctx->stream, b->d_planes, b->d_a_off, b->d_a_len,
                       b->d_b_off, b->d_b_len, d_work, count, logL, d_scratch, d_sc_off, d_dist);
    MALCS_HIP(hipGetLastError());
    return MALCS_OK;
}

int malcs_graph_launch_reach(malcs_batch *b, int W, int logL, const uint32_t *d_work, uint32_t count,
                             const uint32_t *d_scratch, const uint64_t *d_sc_off, const uint8_t *d_skip,
                             void *d_events, const uint64_t *d_ev_off, const uint32_t *d_ev_cap, uint32_t *d_ev_cnt)
{
    reach_kernel_t fn = nullptr;
    switch (b->n_planes) {
    case 2: fn = pick_reach<2>(W); break;
    case 3: fn = pick_reach<3>(W); break;
    case 4: fn = pick_reach<4>(W); break;
    case 8: fn = pick_reach<8>(W); break;
    }
    if (!fn) { malcs_set_error("no reach kernel for W=%d planes=%d", W, b->n_planes); return MALCS_ERR_INTERNAL; }
    const uint32_t ppw = 64u >> logL;
    const uint32_t waves = (count + ppw - 1u) / ppw;
    hipLaunchKernelGGL(fn, dim3((waves + 3u) / 4u), dim3(256), 0, b->ctx->stream, b->d_planes, b->d_a_off, b->d_a_len,
                       b->d_b_off, b->d_b_len, d_work, count, logL, d_scratch, d_sc_off, d_skip,
                       (uint4 *)d_events, d_ev_off, d_ev_cap, d_ev_cnt);
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
