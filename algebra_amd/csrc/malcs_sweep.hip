// malcs_sweep.hip -- the level lists of _lcs_nodes exactly as the reference's sweep fills them
// (algebra/lcs/lcs_graph.py:238-348), including the runs that lie on no minimal alignment.
//
// The graph kernels (malcs_graph.hip) extract the runs _build_graph uses; this kernel is for callers of the
// semi-public _lcs_nodes itself (the reference's tests/lcs/test_lcs_graph.py:20-51 pin its full output).
//
// What the reference emits.  `diagonals[k]` holds 1 + the furthest index reached on diagonal k; expand(k) in
// iteration `it` (:267-315) first re-walks the cells between the previous furthest point of k and the point the
// neighbours now allow to enter at (:287-299), then slides along matches (:305-309), and appends every maximal
// run of matches it passes to the level of the run's last symbol.  Iteration `it` on diagonal k has the budget
// d = |delta| + 2 it - |k - delta| (:294), and the values it reads from k-1 and k+1 are those of budget d - 1:
// the sweep is the furthest-reaching recurrence FR_d[k] = slide(max(FR_{d-1}[k-1], FR_{d-1}[k+1] + 1)) evaluated in
// another order.  So the emitted set is: for every (k, d) with |k| <= d <= D - |k - delta| and d = k (mod 2), the
// maximal match runs on diagonal k between FR_{d-2}[k] + 1 and FR_d[k], with level (row_end + col_end - d)/2 - 1.
// Here all diagonals of one budget are walked at once (they are independent), budgets ascending; the order inside
// a level is restored on the host from the closed-form sweep key (malcs_build.cpp).
//
// Mapping: one workgroup per pair, one lane per diagonal of the current budget, FR in LDS (one array: budgets of
// one parity touch diagonals of one parity) or, beyond LDS_DIAGS diagonals, in HBM scratch.  A lane compares 32
// cells per step: bit planes of both sequences funnel-shifted to the cell, XORed, ORed over the planes.
// Two passes (count, then emit at exact offsets): the number of off-path runs is not bounded by the distance.
#include "malcs_internal.hpp"
#include <algorithm>

int malcs_graph_launch_compact(malcs_ctx *ctx, const void *d_events, const uint64_t *d_ev_off, const uint32_t *d_ev_n,
                               const uint64_t *d_out_off, uint32_t count, void *d_out);

int malcs_sweep_pairs_per_wave(int64_t D);

namespace {

constexpr int LDS_DIAGS = 8192;

template <int NP>
__device__ __forceinline__ uint32_t mismatch32(const uint32_t *__restrict__ A, uint32_t na, const uint32_t *__restrict__ B,
                                               uint32_t nb, uint32_t row, uint32_t col)
{
    const uint32_t ja = row >> 5, sa = row & 31u, jb = col >> 5, sb = col & 31u;
    uint32_t mism = 0;
#pragma unroll
    for (int p = 0; p < NP; p++) {
        const uint32_t alo = ja < na ? A[ja * NP + p] : 0u, ahi = ja + 1u < na ? A[(ja + 1u) * NP + p] : 0u;
        const uint32_t blo = jb < nb ? B[jb * NP + p] : 0u, bhi = jb + 1u < nb ? B[(jb + 1u) * NP + p] : 0u;
        const uint32_t a = (uint32_t)((((uint64_t)ahi << 32) | alo) >> sa);
        const uint32_t b = (uint32_t)((((uint64_t)bhi << 32) | blo) >> sb);
        mism |= a ^ b;
    }
    return mism;
}

template <int NP, bool EMIT>
__global__ void __launch_bounds__(256)
k_lcs_sweep(const uint32_t *__restrict__ planes, const uint32_t *__restrict__ a_off, const uint32_t *__restrict__ a_len,
            const uint32_t *__restrict__ b_off, const uint32_t *__restrict__ b_len, const uint32_t *__restrict__ work,
            const int32_t *__restrict__ dist, int32_t *__restrict__ fr_scratch, const uint64_t *__restrict__ fr_off,
            uint32_t *__restrict__ counts, uint4 *__restrict__ out, const uint64_t *__restrict__ out_off,
            int32_t *__restrict__ reached)
{
    __shared__ int32_t lds_fr[LDS_DIAGS];
    __shared__ uint32_t lds_cnt;
    const uint32_t w = blockIdx.x, pid = work[w];
    const int32_t m = (int32_t)a_len[pid], n = (int32_t)b_len[pid], D = dist[pid], delta = n - m;
    const uint32_t *A = planes + a_off[pid], *B = planes + b_off[pid];
    const uint32_t na = ((uint32_t)m + 31u) >> 5, nb = ((uint32_t)n + 31u) >> 5;
    const int32_t kmin = -((D - delta) / 2), kmax = (D + delta) / 2;          // D >= |delta|, D = delta (mod 2)
    const int32_t ndiag = kmax - kmin + 3;                                    // one guard on either side
    int32_t *FR = ndiag <= LDS_DIAGS ? lds_fr : fr_scratch + fr_off[w];
    // never visited: the reference's diagonals[] entry 0 = "furthest index -1" = row max(0, -k) - 1
    for (int32_t i = (int32_t)threadIdx.x; i < ndiag; i += (int32_t)blockDim.x) {
        const int32_t k = kmin - 1 + i;
        FR[i] = (k < 0 ? -k : 0) - 1;
    }
    if (threadIdx.x == 0) lds_cnt = 0;
    __syncthreads();
    uint32_t mine = 0;
    uint4 *dst = EMIT ? out + out_off[w] : nullptr;
    for (int32_t d = 0; d <= D; d++) {
        const int32_t klo = max(-d, delta - (D - d)), khi = min(d, delta + (D - d));
        for (int32_t k = klo + 2 * (int32_t)threadIdx.x; k <= khi; k += 2 * (int32_t)blockDim.x) {
            const int32_t idx = k - kmin + 1;
            const int32_t prev = FR[idx];
            const int32_t limit = min(m, n - k);                              // rows of diagonal k end here
            if (prev >= limit) continue;                                      // already at the edge of the matrix
            int32_t x = max(FR[idx - 1], FR[idx + 1] + 1);                    // row the neighbours allow to enter at
            x = min(max(x, prev + 1), limit);
            int32_t pos = prev + 1, run_start = 0, stop;
            uint32_t in_run = 0;
            for (;;) {
                uint32_t mism = mismatch32<NP>(A, na, B, nb, (uint32_t)pos, (uint32_t)(pos + k));
                const int32_t valid = limit - pos;
                if (valid < 32) mism |= valid <= 0 ? ~0u : (~0u << valid);
                const int32_t walked = x - pos;                               // cells before x are passed whatever they hold
                const uint32_t stops = walked >= 32 ? 0u : (walked <= 0 ? mism : (mism & (~0u << walked)));
                const int32_t nbits = stops ? __builtin_ctz(stops) : 32;
                const uint32_t match = ~mism & (nbits == 32 ? ~0u : ((1u << nbits) - 1u));
                uint32_t turn = match ^ ((match << 1) | in_run);
                while (turn) {
                    const int32_t i = __builtin_ctz(turn);
                    turn &= turn - 1u;
                    if ((match >> i) & 1u) {
                        run_start = pos + i;
                    } else {
                        const int32_t end = pos + i;                          // first row after the run
                        if (EMIT) {
                            const uint32_t slot = atomicAdd(&lds_cnt, 1u);
                            dst[slot] = make_uint4((uint32_t)run_start, (uint32_t)(run_start + k), (uint32_t)(end - run_start),
                                                   (uint32_t)((2 * end + k - d) / 2 - 1));
                        } else {
                            mine++;
                        }
                    }
                }
                in_run = match >> 31;
                if (nbits < 32) { stop = pos + nbits; break; }
                pos += 32;
            }
            FR[idx] = stop;
            if (d == D && k == delta && !EMIT) reached[w] = stop;
        }
        __syncthreads();
    }
    if (!EMIT) {
        if (mine) atomicAdd(&lds_cnt, mine);
        __syncthreads();
        if (threadIdx.x == 0) counts[w] = lds_cnt;
    }
}


// ---- output-sensitive path for long, similar pairs ---------------------------------------------------------
// The bit-vector kernels of malcs_graph.hip cost (columns x band) whatever the sequences hold; a pair of 1 Mb
// windows a few dozen edits apart is one wave crawling along a million columns.  The furthest-reaching recurrence
// costs O(n / 32 + D^2): slides compare 32 cells per step, and a slide that does not end within a lane's two (one-wave kernels: eight) words is
// finished by the whole workgroup (every thread one word, first mismatch by an LDS minimum).
//
//   k_sweep_reach<NP, REV>   FR_d[k] for d = 0, 1, ... until the corner is reached (=> the distance, without
//                            knowing it beforehand) or `cap` is exceeded.  REV walks the reversed sequences, i.e.
//                            from the sink backwards, and with `tab` keeps every FR_d[k] (row d at d(d+1)/2,
//                            entry (k + d) / 2): the backward reach table.
//   k_sweep_runs<NP>         forward again, now knowing D and the table: on diagonal k at budget d the corners
//                            (prev, FR_d[k]] all have forward cost d; such a corner lies on a minimal alignment
//                            iff the sink reaches it with D - d, i.e. iff its row >= m - FRrev_{D-d}[delta - k].
//                            Emits the maximal match runs that end in such a corner = exactly the runs the
//                            reach kernel of malcs_graph.hip extracts (level = (row_end + col_end - d)/2 - 1).
// Two shapes of workgroup: long or divergent pairs get 512 threads and 8 192 diagonals in LDS, windows below 4 096
// rows at distance <= SMALL_DMAX one wave and 1 024 diagonals -- 5 KB of LDS, so that dozens of pairs share a CU.
constexpr int BIG_THREADS = 512, BIG_DIAGS = 8192, SMALL_THREADS = 64, SMALL_DIAGS = 1024, SMALL_DMAX = 510;
// words (of 32 cells) a lane slides alone before the slide goes to the queue that the whole workgroup finishes: two for
// the workgroups of 512 (a long window's slides are thousands of cells), eight for the one-wave kernels, where a queued
// slide stops every pair of the wave and 1 kb windows at 2 % slide 50 cells on average between edits (C2's sweeps with
// 2 / 4 / 8 words: 0.47 + 0.47, 0.39 + 0.40, 0.37 + 0.39 ms per 100 000)
constexpr int LOCAL_WORDS = 2, SMALL_LOCAL_WORDS = 8;
constexpr int SWEEP_UNROLL = 4;
constexpr int32_t NO_STOP = 0x7fffffff;

template <int NP>
__device__ __forceinline__ uint32_t bits_at(const uint32_t *__restrict__ S, uint32_t nblk, int32_t start, int p)
{
    const int32_t j = start >> 5;                                             // floor, also for negative starts
    const uint32_t s = (uint32_t)start & 31u;
    // unconditional loads from clamped blocks (nblk >= 1), so that the loads of several diagonals can be in flight
    const int32_t last = (int32_t)nblk - 1;
    const uint32_t l0 = S[(uint32_t)min(max(j, 0), last) * NP + p], l1 = S[(uint32_t)min(max(j + 1, 0), last) * NP + p];
    const uint32_t lo = (j >= 0 && j <= last) ? l0 : 0u;
    const uint32_t hi = (j + 1 >= 0 && j + 1 <= last) ? l1 : 0u;
    return (uint32_t)((((uint64_t)hi << 32) | lo) >> s);
}

// mismatch bits of the 32 cells (row + i, col + i); REV: of the problem on the reversed sequences, where cell
// (r, c) is the original cell (m - 1 - r, n - 1 - c).  Cells outside the matrix come out arbitrary.
// DNA: both planes of a block in one 8-byte load (half the loads, clamps and selects of two calls of bits_at)
__device__ __forceinline__ uint2 bits_at_dna(const uint32_t *__restrict__ S, uint32_t nblk, int32_t start)
{
    const int32_t j = start >> 5, last = (int32_t)nblk - 1;
    const uint32_t s = (uint32_t)start & 31u;
    const uint2 l0 = *reinterpret_cast<const uint2 *>(S + (uint32_t)min(max(j, 0), last) * 2u);
    const uint2 l1 = *reinterpret_cast<const uint2 *>(S + (uint32_t)min(max(j + 1, 0), last) * 2u);
    const bool in0 = j >= 0 && j <= last, in1 = j + 1 >= 0 && j + 1 <= last;
    return make_uint2(__builtin_amdgcn_alignbit(in1 ? l1.x : 0u, in0 ? l0.x : 0u, s), __builtin_amdgcn_alignbit(in1 ? l1.y : 0u, in0 ? l0.y : 0u, s));
}

template <int NP, bool REV>
__device__ __forceinline__ uint32_t cells32(const uint32_t *__restrict__ A, uint32_t na, int32_t m,
                                            const uint32_t *__restrict__ B, uint32_t nb, int32_t n, int32_t row, int32_t col)
{
    uint32_t mism = 0;
    const int32_t sa = REV ? m - 32 - row : row, sb = REV ? n - 32 - col : col;
    if constexpr (NP == 2) {
        const uint2 a = bits_at_dna(A, na, sa), b = bits_at_dna(B, nb, sb);
        mism = (a.x ^ b.x) | (a.y ^ b.y);
        return REV ? __brev(mism) : mism;
    }
#pragma unroll
    for (int p = 0; p < NP; p++) mism |= bits_at<NP>(A, na, sa, p) ^ bits_at<NP>(B, nb, sb, p);
    return REV ? __brev(mism) : mism;
}

template <int TH> struct sweep_queue {
    int32_t n;
    int32_t pos[TH * SWEEP_UNROLL], k[TH * SWEEP_UNROLL], stop[TH * SWEEP_UNROLL];
    int32_t first[3];
};

// the slides the lanes could not finish, one after the other, all threads on each
template <int NP, bool REV, int TH>
__device__ __forceinline__ void finish_slides(sweep_queue<TH> &q, const uint32_t *A, uint32_t na, int32_t m, const uint32_t *B,
                                              uint32_t nb, int32_t n)
{
    const int32_t nq = q.n;
    for (int32_t i = 0; i < nq; i++) {
        const int32_t k = q.k[i], limit = min(m, n - k);
        int32_t pos = q.pos[i];
        if (threadIdx.x == 0) q.first[0] = NO_STOP;
        __syncthreads();
        // round r collects in slot r % 3 while thread 0 clears slot (r + 1) % 3; slot (r + 2) % 3 may still be read
        for (uint32_t round = 0;; round++) {
            if (threadIdx.x == 0) q.first[(round + 1u) % 3u] = NO_STOP;
            const int32_t p = pos + 32 * (int32_t)threadIdx.x;
            if (p < limit) {
                uint32_t mism = cells32<NP, REV>(A, na, m, B, nb, n, p, p + k);
                const int32_t valid = limit - p;
                if (valid < 32) mism |= ~0u << valid;
                if (mism) atomicMin(&q.first[round % 3u], p + (int32_t)__builtin_ctz(mism));
            } else if (p - 32 < limit) {
                atomicMin(&q.first[round % 3u], limit);                       // the slide ran into the edge of the matrix
            }
            __syncthreads();
            const int32_t f = q.first[round % 3u];
            if (f != NO_STOP) { if (threadIdx.x == 0) q.stop[i] = f; break; }
            pos += 32 * TH;
        }
        __syncthreads();                                                      // everyone has read the slot before it is reused
    }
}

template <int NP, bool REV, int TH, int DG>
__global__ void __launch_bounds__(TH)
k_sweep_reach(const uint32_t *__restrict__ planes, const uint32_t *__restrict__ a_off, const uint32_t *__restrict__ a_len,
              const uint32_t *__restrict__ b_off, const uint32_t *__restrict__ b_len, const uint32_t *__restrict__ work,
              const int32_t *__restrict__ caps, int32_t *__restrict__ fr_scratch, const uint64_t *__restrict__ fr_off,
              int32_t *__restrict__ tab, const uint64_t *__restrict__ tab_off, int32_t *__restrict__ dist_out)
{
    __shared__ int32_t lds_fr[DG];
    __shared__ sweep_queue<TH> q;
    constexpr int U = TH >= 256 ? SWEEP_UNROLL : 1;      // a single wave rarely has more diagonals than lanes
    constexpr int LW = TH >= 256 ? LOCAL_WORDS : SMALL_LOCAL_WORDS;
    const uint32_t w = blockIdx.x, pid = work[w];
    const int32_t m = (int32_t)a_len[pid], n = (int32_t)b_len[pid], delta = n - m, cap = caps[w];
    const uint32_t *A = planes + a_off[pid], *B = planes + b_off[pid];
    const uint32_t na = ((uint32_t)m + 31u) >> 5, nb = ((uint32_t)n + 31u) >> 5;
    const int32_t ndiag = 2 * cap + 3;                                        // k in [-cap - 1, cap + 1]
    int32_t *FR = ndiag <= DG ? lds_fr : fr_scratch + fr_off[w];
    int32_t *T = tab ? tab + tab_off[w] : nullptr;
    for (int32_t i = (int32_t)threadIdx.x; i < ndiag; i += TH) {
        const int32_t k = i - cap - 1;
        FR[i] = (k < 0 ? -k : 0) - 1;
    }
    if (threadIdx.x == 0) { q.n = 0; q.first[0] = NO_STOP; }
    __syncthreads();
    int32_t found = -1;
    const bool corner_on_band = delta >= -cap && delta <= cap;
    for (int32_t d = 0; d <= cap && corner_on_band; d++) {
        int32_t klo = max(-d, -m), khi = min(d, n);
        if ((klo + d) & 1) klo++;
        if ((khi + d) & 1) khi--;
        const int32_t nk = khi >= klo ? (khi - klo) / 2 + 1 : 0;
        // a thread takes U diagonals of a chunk: their loads are independent and overlap
        for (int32_t base = 0; base < nk; base += TH * U) {
            int32_t stop[U], pos[U], slot[U];
            uint32_t mism[U];
            bool mine[U], open[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int32_t i = base + u * TH + (int32_t)threadIdx.x;
                mine[u] = i < nk;
                const int32_t k = klo + 2 * (mine[u] ? i : 0), idx = k + cap + 1;
                const int32_t prev = FR[idx], limit = min(m, n - k);
                stop[u] = prev;
                slot[u] = -1;
                open[u] = mine[u] && prev < limit;
                pos[u] = min(max(max(FR[idx - 1], FR[idx + 1] + 1), prev + 1), limit);
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int32_t i = base + u * TH + (int32_t)threadIdx.x;
                const int32_t k = klo + 2 * (mine[u] ? i : 0);
                mism[u] = cells32<NP, REV>(A, na, m, B, nb, n, pos[u], pos[u] + k);
            }
            bool queued = false;
#pragma unroll
            for (int u = 0; u < U; u++) {
                if (!open[u]) continue;
                const int32_t i = base + u * TH + (int32_t)threadIdx.x;
                const int32_t k = klo + 2 * i, limit = min(m, n - k);
                uint32_t mm = mism[u];
                int32_t p = pos[u], st = NO_STOP;
                for (int wi = 0;; wi++) {
                    const int32_t valid = limit - p;
                    if (valid < 32) mm |= valid <= 0 ? ~0u : (~0u << valid);
                    if (mm) { st = p + (int32_t)__builtin_ctz(mm); break; }
                    p += 32;
                    if (wi + 1 == LW) break;
                    mm = cells32<NP, REV>(A, na, m, B, nb, n, p, p + k);
                }
                stop[u] = st;
                if (st == NO_STOP) { slot[u] = atomicAdd(&q.n, 1); q.pos[slot[u]] = p; q.k[slot[u]] = k; queued = true; }
            }
            if (__syncthreads_or(queued)) {                                   // rare: somebody's slide goes on
                finish_slides<NP, REV, TH>(q, A, na, m, B, nb, n);
                if (threadIdx.x == 0) q.n = 0;                                // everybody has read it (barriers inside)
                __syncthreads();
#pragma unroll
                for (int u = 0; u < U; u++) if (slot[u] >= 0) stop[u] = q.stop[slot[u]];
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                if (!mine[u]) continue;
                const int32_t i = base + u * TH + (int32_t)threadIdx.x;
                const int32_t k = klo + 2 * i;
                FR[k + cap + 1] = stop[u];                                    // own parity: nobody else reads it in this step
                if (T) T[(int64_t)d * (d + 1) / 2 + (k + d) / 2] = stop[u];
            }
        }
        __syncthreads();                                                      // budget d + 1 reads what budget d wrote
        if (FR[delta + cap + 1] >= m) { found = d; break; }                   // the same LDS / HBM word for every thread
    }
    if (threadIdx.x == 0) dist_out[w] = found;
}

template <int NP, int TH, int DG>
__global__ void __launch_bounds__(TH)
k_sweep_runs(const uint32_t *__restrict__ planes, const uint32_t *__restrict__ a_off, const uint32_t *__restrict__ a_len,
             const uint32_t *__restrict__ b_off, const uint32_t *__restrict__ b_len, const uint32_t *__restrict__ work,
             const int32_t *__restrict__ dists, int32_t *__restrict__ fr_scratch, const uint64_t *__restrict__ fr_off,
             const int32_t *__restrict__ tab, const uint64_t *__restrict__ tab_off,
             uint4 *__restrict__ out, const uint64_t *__restrict__ out_off, const uint32_t *__restrict__ out_cap,
             uint32_t *__restrict__ out_cnt)
{
    __shared__ int32_t lds_fr[DG];
    __shared__ sweep_queue<TH> q;
    constexpr int U = TH >= 256 ? SWEEP_UNROLL : 1;      // a single wave rarely has more diagonals than lanes
    constexpr int LW = TH >= 256 ? LOCAL_WORDS : SMALL_LOCAL_WORDS;
    __shared__ uint32_t lds_cnt;
    const uint32_t w = blockIdx.x, pid = work[w];
    const int32_t m = (int32_t)a_len[pid], n = (int32_t)b_len[pid], delta = n - m, D = dists[w];
    const uint32_t *A = planes + a_off[pid], *B = planes + b_off[pid];
    const uint32_t na = ((uint32_t)m + 31u) >> 5, nb = ((uint32_t)n + 31u) >> 5;
    const int32_t kmin = -((D - delta) / 2), kmax = (D + delta) / 2, ndiag = kmax - kmin + 3;
    int32_t *FR = ndiag <= DG ? lds_fr : fr_scratch + fr_off[w];
    const int32_t *T = tab + tab_off[w];
    uint4 *dst = out + out_off[w];
    const uint32_t cap = out_cap[w];
    for (int32_t i = (int32_t)threadIdx.x; i < ndiag; i += TH) {
        const int32_t k = kmin - 1 + i;
        FR[i] = (k < 0 ? -k : 0) - 1;
    }
    if (threadIdx.x == 0) { q.n = 0; q.first[0] = NO_STOP; lds_cnt = 0; }
    __syncthreads();
    auto emit = [&](int32_t start, int32_t end, int32_t k, int32_t d) {
        const uint32_t slot = atomicAdd(&lds_cnt, 1u);
        if (slot < cap) dst[slot] = make_uint4((uint32_t)start, (uint32_t)(start + k), (uint32_t)(end - start), (uint32_t)((2 * end + k - d) / 2 - 1));
    };
    for (int32_t d = 0; d <= D; d++) {
        const int32_t klo = max(-d, delta - (D - d)), khi = min(d, delta + (D - d));
        const int32_t nk = khi >= klo ? (khi - klo) / 2 + 1 : 0;
        // as in k_sweep_reach: U diagonals per thread, their loads (FR, table, first words) in flight together
        for (int32_t base = 0; base < nk; base += TH * U) {
            int32_t stop[U], slot[U], prev[U], x[U], reach[U];
            uint32_t first_word[U];
            bool mine[U], active[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int32_t i = base + u * TH + (int32_t)threadIdx.x;
                mine[u] = i < nk;
                const int32_t k = klo + 2 * (mine[u] ? i : 0), idx = k - kmin + 1;
                const int32_t limit = min(m, n - k), dd = D - d, kk = delta - k;
                prev[u] = FR[idx];
                stop[u] = prev[u];
                slot[u] = -1;
                active[u] = mine[u] && prev[u] < limit;
                x[u] = min(max(max(FR[idx - 1], FR[idx + 1] + 1), prev[u] + 1), limit);
                reach[u] = T[(int64_t)dd * (dd + 1) / 2 + (kk + dd) / 2];         // rows the sink reaches back on this diagonal
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int32_t i = base + u * TH + (int32_t)threadIdx.x;
                const int32_t k = klo + 2 * (mine[u] ? i : 0);
                first_word[u] = cells32<NP, false>(A, na, m, B, nb, n, x[u], x[u] + k);
            }
            bool queued = false;
#pragma unroll
            for (int u = 0; u < U; u++) {
                if (!active[u]) continue;
                const int32_t i = base + u * TH + (int32_t)threadIdx.x;
                const int32_t k = klo + 2 * i, limit = min(m, n - k);
                uint32_t mism = first_word[u];
                int32_t pos = x[u], st = NO_STOP;
                for (int wi = 0;; wi++) {
                    const int32_t valid = limit - pos;
                    if (valid < 32) mism |= valid <= 0 ? ~0u : (~0u << valid);
                    if (mism) { st = pos + (int32_t)__builtin_ctz(mism); break; }
                    pos += 32;
                    if (wi + 1 == LW) break;
                    mism = cells32<NP, false>(A, na, m, B, nb, n, pos, pos + k);
                }
                stop[u] = st;
                if (st == NO_STOP) { slot[u] = atomicAdd(&q.n, 1); q.pos[slot[u]] = pos; q.k[slot[u]] = k; queued = true; }
            }
            if (__syncthreads_or(queued)) {                                   // rare: somebody's slide goes on
                finish_slides<NP, false, TH>(q, A, na, m, B, nb, n);
                if (threadIdx.x == 0) q.n = 0;                                // everybody has read it (barriers inside)
                __syncthreads();
#pragma unroll
                for (int u = 0; u < U; u++) if (slot[u] >= 0) stop[u] = q.stop[slot[u]];
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                if (!mine[u]) continue;
                const int32_t i = base + u * TH + (int32_t)threadIdx.x;
                const int32_t k = klo + 2 * i;
                FR[k - kmin + 1] = stop[u];                                   // own parity: nobody else reads it in this step
                if (!active[u]) continue;
                const int32_t lo = max(prev[u] + 1, m - reach[u]);            // first corner the sink reaches
                if (stop[u] < lo) continue;
                // start of the run that holds cell s0 - 1 (cells before prev + 1 belong to an earlier budget)
                const int32_t s0 = min(lo, x[u]);
                int32_t s = s0;
                while (s > prev[u] + 1) {
                    const int32_t ws = s - 32;
                    uint32_t mism = cells32<NP, false>(A, na, m, B, nb, n, ws, ws + k);
                    const int32_t before = prev[u] + 1 - ws;
                    if (before > 0) mism |= before >= 32 ? ~0u : ((1u << before) - 1u);
                    if (mism) { s = ws + 32 - (int32_t)__builtin_clz(mism); break; }
                    s = ws;
                }
                // the cells before x: runs may end there
                int32_t pos = s, run_start = s;
                uint32_t in_run = 0;
                while (pos < x[u]) {
                    const uint32_t mism = cells32<NP, false>(A, na, m, B, nb, n, pos, pos + k);
                    const int32_t nbits = min(32, x[u] - pos);
                    const uint32_t low = nbits == 32 ? ~0u : ((1u << nbits) - 1u);
                    const uint32_t match = ~mism & low;
                    uint32_t turn = (match ^ ((match << 1) | in_run)) & low;
                    while (turn) {
                        const int32_t t = __builtin_ctz(turn);
                        turn &= turn - 1u;
                        if ((match >> t) & 1u) run_start = pos + t;
                        else if (pos + t >= lo) emit(run_start, pos + t, k, d);
                    }
                    in_run = (match >> (nbits - 1)) & 1u;
                    pos += 32;
                }
                // from x on everything up to `stop` matches
                const int32_t fs = in_run ? run_start : x[u];
                if (stop[u] > fs) emit(fs, stop[u], k, d);
            }
        }
        __syncthreads();                                                      // budget d + 1 reads what budget d wrote
    }
    if (threadIdx.x == 0) out_cnt[w] = lds_cnt;
}

// ---- several pairs per wave -------------------------------------------------------------------------------------
// A budget step of the one-wave kernels above keeps as many lanes busy as the step has diagonals: d / 2 + 1 of 64 at
// budget d, a sixth of the wave for a 1 kb window at distance 20 -- and the kernels are bound by the instructions they
// issue, not by memory.  Here a wave is PPW groups of 64 / PPW lanes and a group does what the wave did above, for a pair
// of its own: own diagonals (its share of the LDS array: the host sends a chunk here only if every pair's diagonals fit
// it), own budget loop (the wave goes on until its last group is done), own counters.  What stays the wave's: the queue
// of slides that outlast a lane's eight words -- all 64 lanes finish them one after the other, with the parameters of the
// pair the slide belongs to.  The arithmetic is that of k_sweep_reach / k_sweep_runs, statement by statement.
template <int NP, bool REV>
__device__ __forceinline__ void finish_slides_multi(sweep_queue<SMALL_THREADS> &q, const int32_t *q_lane, const uint32_t *A, uint32_t na, int32_t m,
                                                    const uint32_t *B, uint32_t nb, int32_t n)
{
    const int32_t nq = q.n;
    for (int32_t i = 0; i < nq; i++) {
        // the pair of this slide: what its lane holds
        const int src = q_lane[i];
        const unsigned long long Ae = __shfl((unsigned long long)reinterpret_cast<uintptr_t>(A), src, 64);
        const unsigned long long Be = __shfl((unsigned long long)reinterpret_cast<uintptr_t>(B), src, 64);
        const uint32_t *Ai = reinterpret_cast<const uint32_t *>((uintptr_t)Ae), *Bi = reinterpret_cast<const uint32_t *>((uintptr_t)Be);
        const uint32_t nai = __shfl(na, src, 64), nbi = __shfl(nb, src, 64);
        const int32_t mi = __shfl(m, src, 64), ni = __shfl(n, src, 64);
        const int32_t k = q.k[i], limit = min(mi, ni - k);
        int32_t pos = q.pos[i];
        if (threadIdx.x == 0) q.first[0] = NO_STOP;
        __syncthreads();
        for (uint32_t round = 0;; round++) {
            if (threadIdx.x == 0) q.first[(round + 1u) % 3u] = NO_STOP;
            const int32_t p = pos + 32 * (int32_t)threadIdx.x;
            if (p < limit) {
                uint32_t mism = cells32<NP, REV>(Ai, nai, mi, Bi, nbi, ni, p, p + k);
                const int32_t valid = limit - p;
                if (valid < 32) mism |= ~0u << valid;
                if (mism) atomicMin(&q.first[round % 3u], p + (int32_t)__builtin_ctz(mism));
            } else if (p - 32 < limit) {
                atomicMin(&q.first[round % 3u], limit);
            }
            __syncthreads();
            const int32_t f = q.first[round % 3u];
            if (f != NO_STOP) { if (threadIdx.x == 0) q.stop[i] = f; break; }
            pos += 32 * SMALL_THREADS;
        }
        __syncthreads();
    }
}

template <int NP, bool REV, int PPW>
__global__ void __launch_bounds__(SMALL_THREADS)
k_sweep_reach_multi(const uint32_t *__restrict__ planes, const uint32_t *__restrict__ a_off, const uint32_t *__restrict__ a_len,
                    const uint32_t *__restrict__ b_off, const uint32_t *__restrict__ b_len, const uint32_t *__restrict__ work, uint32_t count,
                    const int32_t *__restrict__ caps, int32_t *__restrict__ tab, const uint64_t *__restrict__ tab_off, int32_t *__restrict__ dist_out)
{
    constexpr int LG = SMALL_THREADS / PPW, DGG = SMALL_DIAGS / PPW;
    __shared__ int32_t lds_fr[SMALL_DIAGS];
    __shared__ sweep_queue<SMALL_THREADS> q;
    __shared__ int32_t q_lane[SMALL_THREADS * SWEEP_UNROLL];
    const uint32_t g = threadIdx.x / LG, w = blockIdx.x * PPW + g;
    const int32_t lt = (int32_t)(threadIdx.x % LG);
    const bool valid = w < count;
    const uint32_t pid = work[valid ? w : 0u];
    const int32_t m = (int32_t)a_len[pid], n = (int32_t)b_len[pid], delta = n - m, cap = valid ? caps[w] : -1;
    const uint32_t *A = planes + a_off[pid], *B = planes + b_off[pid];
    const uint32_t na = ((uint32_t)m + 31u) >> 5, nb = ((uint32_t)n + 31u) >> 5;
    const int32_t ndiag = 2 * cap + 3;                                        // (at most DGG: the host's condition for this kernel)
    int32_t *FR = lds_fr + g * DGG;
    int32_t *T = tab && valid ? tab + tab_off[w] : nullptr;
    for (int32_t i = lt; i < min(ndiag, DGG); i += LG) {
        const int32_t k = i - cap - 1;
        FR[i] = (k < 0 ? -k : 0) - 1;
    }
    if (threadIdx.x == 0) { q.n = 0; q.first[0] = NO_STOP; }
    __syncthreads();
    int32_t found = -1;
    bool running = valid && ndiag <= DGG && delta >= -cap && delta <= cap && cap >= 0;
    for (int32_t d = 0; __any(running); d++) {
        int32_t klo = max(-d, -m), khi = min(d, n);
        if ((klo + d) & 1) klo++;
        if ((khi + d) & 1) khi--;
        const int32_t nk = running && khi >= klo ? (khi - klo) / 2 + 1 : 0;
        for (int32_t base = 0; __any(base < nk); base += LG) {
            const int32_t i = base + lt;
            const bool mine = i < nk;
            const int32_t k = klo + 2 * (mine ? i : 0), idx = k + cap + 1;
            const int32_t limit = min(m, n - k);
            int32_t stop = 0, pos = 0, slot = -1;
            bool open = false;
            if (nk > 0) {                                                     // (a group that is done reads nothing)
                const int32_t prev = FR[idx];
                stop = prev;
                open = mine && prev < limit;
                pos = min(max(max(FR[idx - 1], FR[idx + 1] + 1), prev + 1), limit);
            }
            bool queued = false;
            if (open) {
                uint32_t mm = cells32<NP, REV>(A, na, m, B, nb, n, pos, pos + k);
                int32_t p = pos, st = NO_STOP;
                for (int wi = 0;; wi++) {
                    const int32_t left = limit - p;
                    if (left < 32) mm |= left <= 0 ? ~0u : (~0u << left);
                    if (mm) { st = p + (int32_t)__builtin_ctz(mm); break; }
                    p += 32;
                    if (wi + 1 == SMALL_LOCAL_WORDS) break;
                    mm = cells32<NP, REV>(A, na, m, B, nb, n, p, p + k);
                }
                stop = st;
                if (st == NO_STOP) { slot = atomicAdd(&q.n, 1); q.pos[slot] = p; q.k[slot] = k; q_lane[slot] = (int32_t)threadIdx.x; queued = true; }
            }
            if (__syncthreads_or(queued)) {
                finish_slides_multi<NP, REV>(q, q_lane, A, na, m, B, nb, n);
                if (threadIdx.x == 0) q.n = 0;
                __syncthreads();
                if (slot >= 0) stop = q.stop[slot];
            }
            if (mine) {
                FR[idx] = stop;                                               // own parity: nobody else reads it in this step
                if (T) T[(int64_t)d * (d + 1) / 2 + (k + d) / 2] = stop;
            }
        }
        __syncthreads();                                                      // budget d + 1 reads what budget d wrote
        if (running) {
            if (FR[delta + cap + 1] >= m) { found = d; running = false; }
            else if (d + 1 > cap) running = false;
        }
    }
    if (valid && lt == 0) dist_out[w] = found;
}

template <int NP, int PPW>
__global__ void __launch_bounds__(SMALL_THREADS)
k_sweep_runs_multi(const uint32_t *__restrict__ planes, const uint32_t *__restrict__ a_off, const uint32_t *__restrict__ a_len,
                   const uint32_t *__restrict__ b_off, const uint32_t *__restrict__ b_len, const uint32_t *__restrict__ work, uint32_t count,
                   const int32_t *__restrict__ dists, const int32_t *__restrict__ tab, const uint64_t *__restrict__ tab_off,
                   uint4 *__restrict__ out, const uint64_t *__restrict__ out_off, const uint32_t *__restrict__ out_cap,
                   uint32_t *__restrict__ out_cnt)
{
    constexpr int LG = SMALL_THREADS / PPW, DGG = SMALL_DIAGS / PPW;
    __shared__ int32_t lds_fr[SMALL_DIAGS];
    __shared__ sweep_queue<SMALL_THREADS> q;
    __shared__ int32_t q_lane[SMALL_THREADS * SWEEP_UNROLL];
    __shared__ uint32_t lds_cnt[PPW];
    const uint32_t g = threadIdx.x / LG, w = blockIdx.x * PPW + g;
    const int32_t lt = (int32_t)(threadIdx.x % LG);
    const bool valid = w < count;
    const uint32_t wv = valid ? w : 0u, pid = work[wv];
    const int32_t m = (int32_t)a_len[pid], n = (int32_t)b_len[pid], delta = n - m, D = valid ? dists[w] : -1;
    const uint32_t *A = planes + a_off[pid], *B = planes + b_off[pid];
    const uint32_t na = ((uint32_t)m + 31u) >> 5, nb = ((uint32_t)n + 31u) >> 5;
    const int32_t kmin = -((D - delta) / 2), kmax = (D + delta) / 2, ndiag = kmax - kmin + 3;
    int32_t *FR = lds_fr + g * DGG;                                           // (ndiag <= DGG: the host's condition for this kernel)
    const int32_t *T = tab + tab_off[wv];
    uint4 *dst = out + out_off[wv];
    const uint32_t cap = valid ? out_cap[w] : 0u;
    bool running = valid && D >= 0 && ndiag <= DGG;
    if (running)
        for (int32_t i = lt; i < ndiag; i += LG) {
            const int32_t k = kmin - 1 + i;
            FR[i] = (k < 0 ? -k : 0) - 1;
        }
    if (threadIdx.x == 0) { q.n = 0; q.first[0] = NO_STOP; }
    if (lt == 0) lds_cnt[g] = 0;
    __syncthreads();
    auto emit = [&](int32_t start, int32_t end, int32_t k, int32_t d) {
        const uint32_t slot = atomicAdd(&lds_cnt[g], 1u);
        if (slot < cap) dst[slot] = make_uint4((uint32_t)start, (uint32_t)(start + k), (uint32_t)(end - start), (uint32_t)((2 * end + k - d) / 2 - 1));
    };
    for (int32_t d = 0; __any(running); d++) {
        const int32_t klo = max(-d, delta - (D - d)), khi = min(d, delta + (D - d));
        const int32_t nk = running && khi >= klo ? (khi - klo) / 2 + 1 : 0;
        for (int32_t base = 0; __any(base < nk); base += LG) {
            const int32_t i = base + lt;
            const bool mine = i < nk;
            const int32_t k = klo + 2 * (mine ? i : 0), idx = k - kmin + 1;
            const int32_t limit = min(m, n - k), dd = D - d, kk = delta - k;
            int32_t prev = 0, stop = 0, slot = -1, x = 0, reach = 0;
            bool active = false;
            if (nk > 0) {
                prev = FR[idx];
                stop = prev;
                active = mine && prev < limit;
                x = min(max(max(FR[idx - 1], FR[idx + 1] + 1), prev + 1), limit);
                reach = T[(int64_t)dd * (dd + 1) / 2 + (kk + dd) / 2];            // rows the sink reaches back on this diagonal
            }
            bool queued = false;
            if (active) {
                uint32_t mism = cells32<NP, false>(A, na, m, B, nb, n, x, x + k);
                int32_t pos = x, st = NO_STOP;
                for (int wi = 0;; wi++) {
                    const int32_t left = limit - pos;
                    if (left < 32) mism |= left <= 0 ? ~0u : (~0u << left);
                    if (mism) { st = pos + (int32_t)__builtin_ctz(mism); break; }
                    pos += 32;
                    if (wi + 1 == SMALL_LOCAL_WORDS) break;
                    mism = cells32<NP, false>(A, na, m, B, nb, n, pos, pos + k);
                }
                stop = st;
                if (st == NO_STOP) { slot = atomicAdd(&q.n, 1); q.pos[slot] = pos; q.k[slot] = k; q_lane[slot] = (int32_t)threadIdx.x; queued = true; }
            }
            if (__syncthreads_or(queued)) {
                finish_slides_multi<NP, false>(q, q_lane, A, na, m, B, nb, n);
                if (threadIdx.x == 0) q.n = 0;
                __syncthreads();
                if (slot >= 0) stop = q.stop[slot];
            }
            if (mine) FR[idx] = stop;                                         // own parity: nobody else reads it in this step
            const int32_t lo = max(prev + 1, m - reach);                      // first corner the sink reaches
            if (!active || stop < lo) continue;                               // (the wave meets again at the loop's condition)
            const int32_t s0 = min(lo, x);
            int32_t s = s0;
            while (s > prev + 1) {
                const int32_t ws = s - 32;
                uint32_t mism = cells32<NP, false>(A, na, m, B, nb, n, ws, ws + k);
                const int32_t before = prev + 1 - ws;
                if (before > 0) mism |= before >= 32 ? ~0u : ((1u << before) - 1u);
                if (mism) { s = ws + 32 - (int32_t)__builtin_clz(mism); break; }
                s = ws;
            }
            int32_t pos = s, run_start = s;
            uint32_t in_run = 0;
            while (pos < x) {
                const uint32_t mism = cells32<NP, false>(A, na, m, B, nb, n, pos, pos + k);
                const int32_t nbits = min(32, x - pos);
                const uint32_t low = nbits == 32 ? ~0u : ((1u << nbits) - 1u);
                const uint32_t match = ~mism & low;
                uint32_t turn = (match ^ ((match << 1) | in_run)) & low;
                while (turn) {
                    const int32_t t = __builtin_ctz(turn);
                    turn &= turn - 1u;
                    if ((match >> t) & 1u) run_start = pos + t;
                    else if (pos + t >= lo) emit(run_start, pos + t, k, d);
                }
                in_run = (match >> (nbits - 1)) & 1u;
                pos += 32;
            }
            const int32_t fs = in_run ? run_start : x;
            if (stop > fs) emit(fs, stop, k, d);
        }
        __syncthreads();                                                      // budget d + 1 reads what budget d wrote
        if (running && d + 1 > D) running = false;
    }
    if (valid && lt == 0) out_cnt[w] = lds_cnt[g];
}

// device memory of a call, from the context's pool (a hipMalloc costs a fraction of a millisecond and a hipFree waits for
// the device; a sweep over a C2 batch used to make fourteen of each)
template <typename T> struct dbuf {
    malcs_ctx *c;
    T *p = nullptr;
    explicit dbuf(malcs_ctx *ctx) : c(ctx) {}
    dbuf(const dbuf &) = delete;
    dbuf &operator=(const dbuf &) = delete;
    ~dbuf() { release(); }
    void release() { if (p) { malcs_pool_free(c, p); p = nullptr; } }
    int alloc(size_t n) { release(); return malcs_pool_alloc(c, (void **)&p, (n ? n : 1) * sizeof(T)) == hipSuccess ? 0 : -1; }
};

template <int NP>
int launch(malcs_batch *b, bool emit, uint32_t count, const uint32_t *d_work, int32_t *d_fr, const uint64_t *d_fr_off,
           uint32_t *d_counts, uint4 *d_out, const uint64_t *d_out_off, int32_t *d_reached)
{
    if (emit) hipLaunchKernelGGL((k_lcs_sweep<NP, true>), dim3(count), dim3(256), 0, b->ctx->stream, b->d_planes, b->d_a_off,
                                 b->d_a_len, b->d_b_off, b->d_b_len, d_work, b->d_dist, d_fr, d_fr_off, d_counts, d_out, d_out_off, d_reached);
    else hipLaunchKernelGGL((k_lcs_sweep<NP, false>), dim3(count), dim3(256), 0, b->ctx->stream, b->d_planes, b->d_a_off,
                            b->d_a_len, b->d_b_off, b->d_b_len, d_work, b->d_dist, d_fr, d_fr_off, d_counts, d_out, d_out_off, d_reached);
    MALCS_HIP(hipGetLastError());
    return MALCS_OK;
}

int launch_any(malcs_batch *b, bool emit, uint32_t count, const uint32_t *d_work, int32_t *d_fr, const uint64_t *d_fr_off,
               uint32_t *d_counts, uint4 *d_out, const uint64_t *d_out_off, int32_t *d_reached)
{
    switch (b->n_planes) {
    case 2: return launch<2>(b, emit, count, d_work, d_fr, d_fr_off, d_counts, d_out, d_out_off, d_reached);
    case 3: return launch<3>(b, emit, count, d_work, d_fr, d_fr_off, d_counts, d_out, d_out_off, d_reached);
    case 4: return launch<4>(b, emit, count, d_work, d_fr, d_fr_off, d_counts, d_out, d_out_off, d_reached);
    case 8: return launch<8>(b, emit, count, d_work, d_fr, d_fr_off, d_counts, d_out, d_out_off, d_reached);
    }
    malcs_set_error("unsupported plane count %d", b->n_planes);
    return MALCS_ERR_ARG;
}

}  // namespace

// Runs of the sweep for the pairs in `work` (ids into the batch; both sequences non-empty; b->d_dist holds their
// distances, `dist` is its host copy).  out_runs receives, pair after pair, {row, col, length, level} in no
// particular order; out_first[i] .. out_first[i + 1] is the slice of work[i].
int malcs_sweep_runs(malcs_batch *b, const std::vector<uint32_t> &work, const int32_t *dist,
                     std::vector<uint32_t> &out_runs, std::vector<uint64_t> &out_first)
{
    hipStream_t st = b->ctx->stream;
    out_first.assign(work.size() + 1, 0);
    out_runs.clear();
    malcs_stage_trace stages("sweep runs");
    size_t free_b = 0, total_b = 0;
    MALCS_HIP(hipMemGetInfo(&free_b, &total_b));
    stages.lap("memory info");
    const uint64_t budget = std::min<uint64_t>(free_b / 4, 16ull << 30);      // bytes of runs per chunk
    size_t pos = 0;
    while (pos < work.size()) {
        // chunk: bounded FR scratch (only pairs with more diagonals than LDS holds need any)
        std::vector<uint64_t> fr_off;
        uint64_t fr_total = 0;
        size_t end = pos;
        while (end < work.size() && end - pos < (1u << 20)) {
            const uint64_t nd = (uint64_t)dist[work[end]] + 3u;
            const uint64_t need = nd > (uint64_t)LDS_DIAGS ? nd : 0;
            if (end > pos && (fr_total + need) * 4ull > budget) break;
            fr_off.push_back(fr_total);
            fr_total += need;
            end++;
        }
        const uint32_t cnt = (uint32_t)(end - pos);
        dbuf<uint32_t> d_work(b->ctx), d_counts(b->ctx);
        dbuf<int32_t> d_fr(b->ctx), d_reached(b->ctx);
        dbuf<uint64_t> d_fr_off(b->ctx), d_out_off(b->ctx);
        dbuf<uint4> d_out(b->ctx);
        if (d_work.alloc(cnt) || d_counts.alloc(cnt) || d_fr.alloc(fr_total) || d_reached.alloc(cnt) || d_fr_off.alloc(cnt) ||
            d_out_off.alloc(cnt)) {
            malcs_set_error("device allocation failed (sweep chunk of %u pairs)", cnt);
            return MALCS_ERR_NOMEM;
        }
        MALCS_HIP(malcs_h2d(st, d_work.p, work.data() + pos, cnt * 4ull));
        MALCS_HIP(malcs_h2d(st, d_fr_off.p, fr_off.data(), cnt * 8ull));
        MALCS_HIP(hipMemsetAsync(d_reached.p, 0xff, cnt * 4ull, st));
        int rc = launch_any(b, false, cnt, d_work.p, d_fr.p, d_fr_off.p, d_counts.p, nullptr, nullptr, d_reached.p);
        if (rc) return rc;
        std::vector<uint32_t> counts(cnt);
        std::vector<int32_t> reached(cnt);
        MALCS_HIP(malcs_d2h(st, counts.data(), d_counts.p, cnt * 4ull));
        MALCS_HIP(malcs_d2h(st, reached.data(), d_reached.p, cnt * 4ull));
        MALCS_HIP(malcs_sync(st));
        std::vector<uint64_t> off(cnt);
        uint64_t total = 0;
        for (uint32_t s = 0; s < cnt; s++) {
            if (reached[s] != (int32_t)b->a_len[work[pos + s]]) {
                malcs_set_error("pair %u: the sweep did not end in the corner at distance %d (internal error)", work[pos + s],
                                dist[work[pos + s]]);
                return MALCS_ERR_INTERNAL;
            }
            off[s] = total;
            total += counts[s];
        }
        if (total * 16ull > free_b / 2) {
            malcs_set_error("the sweep passes %llu runs in this chunk; the device has %llu MiB free",
                            (unsigned long long)total, (unsigned long long)(free_b >> 20));
            return MALCS_ERR_NOMEM;
        }
        if (d_out.alloc(total)) { malcs_set_error("device allocation failed (%llu runs)", (unsigned long long)total); return MALCS_ERR_NOMEM; }
        MALCS_HIP(malcs_h2d(st, d_out_off.p, off.data(), cnt * 8ull));
        rc = launch_any(b, true, cnt, d_work.p, d_fr.p, d_fr_off.p, d_counts.p, d_out.p, d_out_off.p, d_reached.p);
        if (rc) return rc;
        const size_t base = out_runs.size();
        out_runs.resize(base + total * 4);
        if (total) MALCS_HIP(malcs_d2h(st, out_runs.data() + base, d_out.p, total * 16ull));
        MALCS_HIP(malcs_sync(st));
        for (uint32_t s = 0; s < cnt; s++) out_first[pos + s + 1] = base / 4 + off[s] + counts[s];
        pos = end;
    }
    return MALCS_OK;
}

namespace {

template <int NP>
int launch_reach(malcs_batch *b, bool small, bool rev, uint32_t count, const uint32_t *d_work, const int32_t *d_caps, int32_t *d_fr,
                 const uint64_t *d_fr_off, int32_t *d_tab, const uint64_t *d_tab_off, int32_t *d_dist, int ppw = 1)
{
#define SWEEP_REACH_MULTI(REV, PPW)                                                                                         \
    hipLaunchKernelGGL((k_sweep_reach_multi<NP, REV, PPW>), dim3((count + PPW - 1) / PPW), dim3(SMALL_THREADS), 0, b->ctx->stream, b->d_planes, \
                       b->d_a_off, b->d_a_len, b->d_b_off, b->d_b_len, d_work, count, d_caps, d_tab, d_tab_off, d_dist)
    if (small && ppw > 1) {
        if (ppw == 8)      { if (rev) SWEEP_REACH_MULTI(true, 8); else SWEEP_REACH_MULTI(false, 8); }
        else if (ppw == 4) { if (rev) SWEEP_REACH_MULTI(true, 4); else SWEEP_REACH_MULTI(false, 4); }
        else               { if (rev) SWEEP_REACH_MULTI(true, 2); else SWEEP_REACH_MULTI(false, 2); }
        MALCS_HIP(hipGetLastError());
        return MALCS_OK;
    }
#undef SWEEP_REACH_MULTI
#define SWEEP_REACH(REV, TH, DG)                                                                                            \
    hipLaunchKernelGGL((k_sweep_reach<NP, REV, TH, DG>), dim3(count), dim3(TH), 0, b->ctx->stream, b->d_planes, b->d_a_off, b->d_a_len, \
                       b->d_b_off, b->d_b_len, d_work, d_caps, d_fr, d_fr_off, d_tab, d_tab_off, d_dist)
    if (small) { if (rev) SWEEP_REACH(true, SMALL_THREADS, SMALL_DIAGS); else SWEEP_REACH(false, SMALL_THREADS, SMALL_DIAGS); }
    else       { if (rev) SWEEP_REACH(true, BIG_THREADS, BIG_DIAGS);     else SWEEP_REACH(false, BIG_THREADS, BIG_DIAGS); }
#undef SWEEP_REACH
    MALCS_HIP(hipGetLastError());
    return MALCS_OK;
}

int launch_reach_any(malcs_batch *b, bool small, bool rev, uint32_t count, const uint32_t *d_work, const int32_t *d_caps, int32_t *d_fr,
                     const uint64_t *d_fr_off, int32_t *d_tab, const uint64_t *d_tab_off, int32_t *d_dist, int ppw = 1)
{
    switch (b->n_planes) {
    case 2: return launch_reach<2>(b, small, rev, count, d_work, d_caps, d_fr, d_fr_off, d_tab, d_tab_off, d_dist, ppw);
    case 3: return launch_reach<3>(b, small, rev, count, d_work, d_caps, d_fr, d_fr_off, d_tab, d_tab_off, d_dist, ppw);
    case 4: return launch_reach<4>(b, small, rev, count, d_work, d_caps, d_fr, d_fr_off, d_tab, d_tab_off, d_dist, ppw);
    case 8: return launch_reach<8>(b, small, rev, count, d_work, d_caps, d_fr, d_fr_off, d_tab, d_tab_off, d_dist, ppw);
    }
    malcs_set_error("unsupported plane count %d", b->n_planes);
    return MALCS_ERR_ARG;
}

template <int NP>
int launch_runs(malcs_batch *b, bool small, uint32_t count, const uint32_t *d_work, const int32_t *d_dists, int32_t *d_fr, const uint64_t *d_fr_off,
                const int32_t *d_tab, const uint64_t *d_tab_off, uint4 *d_out, const uint64_t *d_out_off, const uint32_t *d_out_cap,
                uint32_t *d_out_cnt, int ppw = 1)
{
    if (small && ppw == 8)
        hipLaunchKernelGGL((k_sweep_runs_multi<NP, 8>), dim3((count + 7) / 8), dim3(SMALL_THREADS), 0, b->ctx->stream, b->d_planes, b->d_a_off,
                           b->d_a_len, b->d_b_off, b->d_b_len, d_work, count, d_dists, d_tab, d_tab_off, d_out, d_out_off, d_out_cap, d_out_cnt);
    else if (small && ppw == 4)
        hipLaunchKernelGGL((k_sweep_runs_multi<NP, 4>), dim3((count + 3) / 4), dim3(SMALL_THREADS), 0, b->ctx->stream, b->d_planes, b->d_a_off,
                           b->d_a_len, b->d_b_off, b->d_b_len, d_work, count, d_dists, d_tab, d_tab_off, d_out, d_out_off, d_out_cap, d_out_cnt);
    else if (small && ppw == 2)
        hipLaunchKernelGGL((k_sweep_runs_multi<NP, 2>), dim3((count + 1) / 2), dim3(SMALL_THREADS), 0, b->ctx->stream, b->d_planes, b->d_a_off,
                           b->d_a_len, b->d_b_off, b->d_b_len, d_work, count, d_dists, d_tab, d_tab_off, d_out, d_out_off, d_out_cap, d_out_cnt);
    else if (small) hipLaunchKernelGGL((k_sweep_runs<NP, SMALL_THREADS, SMALL_DIAGS>), dim3(count), dim3(SMALL_THREADS), 0, b->ctx->stream, b->d_planes,
                                  b->d_a_off, b->d_a_len, b->d_b_off, b->d_b_len, d_work, d_dists, d_fr, d_fr_off, d_tab, d_tab_off, d_out,
                                  d_out_off, d_out_cap, d_out_cnt);
    else hipLaunchKernelGGL((k_sweep_runs<NP, BIG_THREADS, BIG_DIAGS>), dim3(count), dim3(BIG_THREADS), 0, b->ctx->stream, b->d_planes,
                            b->d_a_off, b->d_a_len, b->d_b_off, b->d_b_len, d_work, d_dists, d_fr, d_fr_off, d_tab, d_tab_off, d_out,
                            d_out_off, d_out_cap, d_out_cnt);
    MALCS_HIP(hipGetLastError());
    return MALCS_OK;
}

// pairs per wave of the one-wave sweeps: as many as the widest pair of the chunk leaves room for (its diagonals in its
// group's share of the LDS array); MALCS_SWEEP_PPW caps it (1: the one-pair kernels)
int ppw_of_diagonals(int64_t ndiag)
{
    return ndiag <= SMALL_DIAGS / 8 ? 8 : ndiag <= SMALL_DIAGS / 4 ? 4 : ndiag <= SMALL_DIAGS / 2 ? 2 : 1;
}

int sweep_ppw(bool small, const std::vector<int64_t> &ndiag)
{
    if (!small || ndiag.empty()) return 1;
    int64_t widest = 0;
    for (int64_t v : ndiag) widest = std::max(widest, v);
    int ppw = ppw_of_diagonals(widest);
    if (const char *e = getenv("MALCS_SWEEP_PPW")) ppw = std::min(ppw, atoi(e) >= 8 ? 8 : atoi(e) >= 4 ? 4 : atoi(e) >= 2 ? 2 : 1);
    return ppw;
}

// scratch for FR: pairs whose diagonals do not fit LDS
uint64_t fr_layout(const std::vector<int64_t> &ndiag, int lds_diags, std::vector<uint64_t> &off)
{
    uint64_t total = 0;
    off.resize(ndiag.size());
    for (size_t i = 0; i < ndiag.size(); i++) { off[i] = total; if (ndiag[i] > lds_diags) total += (uint64_t)ndiag[i]; }
    return total;
}

}  // namespace

// Distances of long pairs without a bound known beforehand: forward sweep until the corner, at most `cap` edits.
// out[i] = distance of pairs[i], or -1 when it exceeds cap.  Synchronous.
int malcs_sweep_distances(malcs_batch *b, const std::vector<uint32_t> &pairs, int32_t cap, std::vector<int32_t> &out)
{
    hipStream_t st = b->ctx->stream;
    const uint32_t cnt = (uint32_t)pairs.size();
    out.assign(cnt, -1);
    if (!cnt) return MALCS_OK;
    std::vector<int32_t> caps(cnt);
    std::vector<int64_t> ndiag(cnt);
    for (uint32_t i = 0; i < cnt; i++) {
        caps[i] = (int32_t)std::min<int64_t>(cap, (int64_t)b->a_len[pairs[i]] + b->b_len[pairs[i]]);
        ndiag[i] = 2 * (int64_t)caps[i] + 3;
    }
    std::vector<uint64_t> fr_off;
    const uint64_t fr_total = fr_layout(ndiag, BIG_DIAGS, fr_off);
    dbuf<uint32_t> d_work(b->ctx);
    dbuf<int32_t> d_caps(b->ctx), d_fr(b->ctx), d_dist(b->ctx);
    dbuf<uint64_t> d_fr_off(b->ctx);
    if (d_work.alloc(cnt) || d_caps.alloc(cnt) || d_fr.alloc(fr_total) || d_dist.alloc(cnt) || d_fr_off.alloc(cnt)) {
        malcs_set_error("device allocation failed (sweep over %u pairs)", cnt);
        return MALCS_ERR_NOMEM;
    }
    MALCS_HIP(malcs_h2d(st, d_work.p, pairs.data(), cnt * 4ull));
    MALCS_HIP(malcs_h2d(st, d_caps.p, caps.data(), cnt * 4ull));
    MALCS_HIP(malcs_h2d(st, d_fr_off.p, fr_off.data(), cnt * 8ull));
    int rc = launch_reach_any(b, false, false, cnt, d_work.p, d_caps.p, d_fr.p, d_fr_off.p, nullptr, nullptr, d_dist.p);
    if (rc) return rc;
    MALCS_HIP(malcs_d2h(st, out.data(), d_dist.p, cnt * 4ull));
    MALCS_HIP(malcs_sync(st));
    return MALCS_OK;
}

// Runs on the minimal alignments of the pairs in `work` (distances known: dist[pid]) by the two sweeps: backward
// with the reach table, forward with emission.  Output as malcs_sweep_runs.  ms[0] += backward, ms[1] += forward pass.
// With unknown_cap > 0 the distances are not known yet: the backward sweep runs up to that many edits, writes what
// it finds to dist[pid] (-1: more than unknown_cap, no runs for that pair) and sizes its table for the cap.
int malcs_sweep_optimal_runs(malcs_batch *b, bool small, const std::vector<uint32_t> &work, int32_t *dist, int32_t unknown_cap,
                             pinned_vec<uint32_t> &out_runs, std::vector<uint64_t> &out_first, float *ms,
                             malcs_sweep_hook hook, void *hook_user)
{
    const int lds_diags = small ? SMALL_DIAGS : BIG_DIAGS;
    hipStream_t st = b->ctx->stream;
    malcs_ctx *ctx = b->ctx;
    out_first.assign(work.size() + 1, 0);
    out_runs.clear();
    malcs_stage_trace stages("sweep runs");
    size_t free_b = 0, total_b = 0;
    MALCS_HIP(hipMemGetInfo(&free_b, &total_b));
    stages.lap("memory info");
    const uint64_t budget = std::min<uint64_t>(free_b / 2, 64ull << 30) / 4;  // ints of table per chunk
    size_t pos = 0;
    while (pos < work.size()) {
        std::vector<uint64_t> tab_off;
        std::vector<int32_t> dists;
        std::vector<int64_t> nd_back, nd_fwd;
        {
            const size_t guess = std::min<size_t>(work.size() - pos, (size_t)1 << 20);
            tab_off.reserve(guess); dists.reserve(guess); nd_back.reserve(guess); nd_fwd.reserve(guess);
        }
        uint64_t tab_total = 0;
        size_t end = pos;
        while (end < work.size()) {
            const int64_t D = unknown_cap ? std::min<int64_t>(unknown_cap, (int64_t)b->a_len[work[end]] + b->b_len[work[end]])
                                          : dist[work[end]];
            const uint64_t need = (uint64_t)(D + 1) * (uint64_t)(D + 2) / 2;
            if (end > pos && tab_total + need > budget) break;
            // a chunk of one-wave sweeps holds pairs of one width: its widest pair decides how many pairs share a wave
            if (small && end > pos && malcs_sweep_pairs_per_wave(D) != malcs_sweep_pairs_per_wave(dists[0])) break;
            if (need > budget) {
                malcs_set_error("pair %u at distance %lld needs %llu MiB of reach table; the device has %llu MiB free", work[end],
                                (long long)D, (unsigned long long)(need >> 18), (unsigned long long)(free_b >> 20));
                return MALCS_ERR_NOMEM;
            }
            tab_off.push_back(tab_total); tab_total += need; dists.push_back((int32_t)D);
            nd_back.push_back(2 * D + 3); nd_fwd.push_back(D + 3);
            end++;
        }
        const uint32_t cnt = (uint32_t)(end - pos);
        std::vector<uint64_t> fr_off_b, fr_off_f;
        const uint64_t fr_b = fr_layout(nd_back, lds_diags, fr_off_b), fr_f = fr_layout(nd_fwd, lds_diags, fr_off_f);
        dbuf<uint32_t> d_work(b->ctx), d_cap(b->ctx), d_cnt(b->ctx);
        dbuf<int32_t> d_dists(b->ctx), d_fr(b->ctx), d_tab(b->ctx), d_found(b->ctx);
        dbuf<uint64_t> d_fr_off(b->ctx), d_tab_off(b->ctx), d_out_off(b->ctx);
        dbuf<uint4> d_out(b->ctx);
        if (d_work.alloc(cnt) || d_cap.alloc(cnt) || d_cnt.alloc(cnt) || d_dists.alloc(cnt) || d_fr.alloc(std::max(fr_b, fr_f)) ||
            d_tab.alloc(tab_total) || d_found.alloc(cnt) || d_fr_off.alloc(cnt) || d_tab_off.alloc(cnt) || d_out_off.alloc(cnt)) {
            malcs_set_error("device allocation failed (sweep chunk of %u pairs, %llu MiB of reach table)", cnt, (unsigned long long)(tab_total >> 18));
            return MALCS_ERR_NOMEM;
        }
        stages.lap("chunk, allocations");
        MALCS_HIP(malcs_h2d(st, d_work.p, work.data() + pos, cnt * 4ull));
        MALCS_HIP(malcs_h2d(st, d_dists.p, dists.data(), cnt * 4ull));
        MALCS_HIP(malcs_h2d(st, d_tab_off.p, tab_off.data(), cnt * 8ull));
        if (fr_b) MALCS_HIP(malcs_h2d(st, d_fr_off.p, fr_off_b.data(), cnt * 8ull));   // (read only by pairs whose diagonals do not fit LDS)
        // an entry never written reaches nothing (with a cap instead of the distance the table may be GBs, and every
        // entry the forward pass looks up has been written: skip the fill)
        if (!unknown_cap) MALCS_HIP(hipMemsetAsync(d_tab.p, 0xff, tab_total * 4ull, st));
        MALCS_HIP(hipEventRecord(ctx->ev0, st));
        int rc = launch_reach_any(b, small, true, cnt, d_work.p, d_dists.p, d_fr.p, d_fr_off.p, d_tab.p, d_tab_off.p, d_found.p,
                                  sweep_ppw(small, nd_back));
        if (rc) return rc;
        MALCS_HIP(hipEventRecord(ctx->ev1, st));
        std::vector<int32_t> found(cnt);
        MALCS_HIP(malcs_d2h(st, found.data(), d_found.p, cnt * 4ull));
        MALCS_HIP(malcs_sync(st));
        float t = 0;
        MALCS_HIP(hipEventElapsedTime(&t, ctx->ev0, ctx->ev1));
        ms[0] += t;
        stages.lap("backward sweep");
        // the pairs that go on to the forward pass (all of them when the distances were known)
        const uint32_t all = cnt;
        std::vector<uint32_t> keep, fwd_work;
        std::vector<uint64_t> fwd_tab_off;
        std::vector<int32_t> fwd_dists;
        uint32_t cntf;
        uint64_t fr_fwd = fr_f;                                               // words of scratch the forward pass's diagonals need beyond LDS
        if (!unknown_cap) {
            // the distances were known: every pair goes on, and its number, distance, table offset and the forward
            // pass's layout (worked out above) are what the device already holds or the host already has
            for (uint32_t s = 0; s < all; s++)
                if (found[s] != dists[s]) {
                    malcs_set_error("pair %u: the backward sweep found distance %d, expected %d (internal error)", work[pos + s], found[s], dists[s]);
                    return MALCS_ERR_INTERNAL;
                }
            cntf = all;
            keep.resize(all);
            for (uint32_t s = 0; s < all; s++) keep[s] = s;
            fwd_work.assign(work.begin() + (std::ptrdiff_t)pos, work.begin() + (std::ptrdiff_t)end);
            fwd_dists = dists;
        } else {
            nd_fwd.clear();
            for (uint32_t s = 0; s < all; s++) {
                const uint32_t pid = work[pos + s];
                dist[pid] = found[s];
                if (found[s] < 0) continue;
                keep.push_back(s); fwd_work.push_back(pid); fwd_tab_off.push_back(tab_off[s]);
                nd_fwd.push_back((int64_t)found[s] + 3);
            }
            cntf = (uint32_t)keep.size();
            if (!cntf) { for (uint32_t s = 0; s < all; s++) out_first[pos + s + 1] = out_runs.size() / 4; pos = end; continue; }
            fwd_dists.resize(cntf);
            for (uint32_t f = 0; f < cntf; f++) fwd_dists[f] = found[keep[f]];
            const uint64_t fr_f2 = fr_layout(nd_fwd, lds_diags, fr_off_f);
            if (fr_f2 > std::max(fr_b, fr_f)) { malcs_set_error("sweep scratch too small (internal error)"); return MALCS_ERR_INTERNAL; }
            fr_fwd = fr_f2;
            MALCS_HIP(malcs_h2d(st, d_work.p, fwd_work.data(), cntf * 4ull));
            MALCS_HIP(malcs_h2d(st, d_dists.p, fwd_dists.data(), cntf * 4ull));
            MALCS_HIP(malcs_h2d(st, d_tab_off.p, fwd_tab_off.data(), cntf * 8ull));
        }
        std::vector<uint32_t> cap(cntf), got(cntf);
        std::vector<uint64_t> off(cntf);
        uint64_t total = 0;
        for (uint32_t f = 0; f < cntf; f++) {
            cap[f] = 64u + 16u * (uint32_t)fwd_dists[f];
            off[f] = total; total += cap[f];
        }
        if (fr_fwd) MALCS_HIP(malcs_h2d(st, d_fr_off.p, fr_off_f.data(), cntf * 8ull));
        const int ppw_f = sweep_ppw(small, nd_fwd);
        for (int attempt = 0;; attempt++) {
            d_out.release();
            if (d_out.alloc(total)) { malcs_set_error("device allocation failed (%llu runs)", (unsigned long long)total); return MALCS_ERR_NOMEM; }
            MALCS_HIP(malcs_h2d(st, d_cap.p, cap.data(), cntf * 4ull));
            MALCS_HIP(malcs_h2d(st, d_out_off.p, off.data(), cntf * 8ull));
            MALCS_HIP(hipEventRecord(ctx->ev1, st));
            switch (b->n_planes) {
            case 2: rc = launch_runs<2>(b, small, cntf, d_work.p, d_dists.p, d_fr.p, d_fr_off.p, d_tab.p, d_tab_off.p, d_out.p, d_out_off.p, d_cap.p, d_cnt.p, ppw_f); break;
            case 3: rc = launch_runs<3>(b, small, cntf, d_work.p, d_dists.p, d_fr.p, d_fr_off.p, d_tab.p, d_tab_off.p, d_out.p, d_out_off.p, d_cap.p, d_cnt.p, ppw_f); break;
            case 4: rc = launch_runs<4>(b, small, cntf, d_work.p, d_dists.p, d_fr.p, d_fr_off.p, d_tab.p, d_tab_off.p, d_out.p, d_out_off.p, d_cap.p, d_cnt.p, ppw_f); break;
            case 8: rc = launch_runs<8>(b, small, cntf, d_work.p, d_dists.p, d_fr.p, d_fr_off.p, d_tab.p, d_tab_off.p, d_out.p, d_out_off.p, d_cap.p, d_cnt.p, ppw_f); break;
            default: malcs_set_error("unsupported plane count %d", b->n_planes); rc = MALCS_ERR_ARG;
            }
            if (rc) return rc;
            MALCS_HIP(hipEventRecord(ctx->ev2, st));
            MALCS_HIP(malcs_d2h(st, got.data(), d_cnt.p, cntf * 4ull));
            MALCS_HIP(malcs_sync(st));
            MALCS_HIP(hipEventElapsedTime(&t, ctx->ev1, ctx->ev2));
            ms[1] += t;
            bool overflow = false;
            for (uint32_t s = 0; s < cntf; s++) if (got[s] > cap[s]) overflow = true;
            if (!overflow) break;
            if (attempt) { malcs_set_error("run buffer overflow persisted after a resize"); return MALCS_ERR_INTERNAL; }
            total = 0;                                                        // the counts are exact: size every region to its count, redo
            for (uint32_t s = 0; s < cntf; s++) { cap[s] = std::max(cap[s], got[s]); off[s] = total; total += cap[s]; }
        }
        stages.lap("forward sweep");
        // gather the used parts on the device, bring them home in one copy
        size_t base = out_runs.size();
        std::vector<uint64_t> packed_off(cntf);
        uint64_t kept = 0;
        for (uint32_t s = 0; s < cntf; s++) { packed_off[s] = kept; kept += got[s]; }
        out_runs.resize(base + kept * 4);
        stages.lap("host vector for the runs");
        if (kept) {
            dbuf<uint4> d_packed(b->ctx);
            dbuf<uint64_t> d_packed_off(b->ctx);
            if (d_packed.alloc(kept) || d_packed_off.alloc(cntf)) { malcs_set_error("device allocation failed (%llu runs)", (unsigned long long)kept); return MALCS_ERR_NOMEM; }
            MALCS_HIP(malcs_h2d(st, d_packed_off.p, packed_off.data(), cntf * 8ull));
            rc = malcs_graph_launch_compact(ctx, d_out.p, d_out_off.p, d_cnt.p, d_packed_off.p, cntf, d_packed.p);
            if (rc) return rc;
            bool runs_home = false;
            if (hook) {
                // the runs of this chunk sit compacted in device memory: whoever wants to go on with them there (the graph
                // kernels of malcs_gbuild.hip, which also leave them sorted) does so before they come home
                std::vector<uint64_t> index(cntf);
                for (uint32_t f = 0; f < cntf; f++) index[f] = pos + keep[f];
                malcs_sweep_chunk chunk{d_packed.p, d_packed_off.p, d_cnt.p, d_work.p, cntf, fwd_work.data(), got.data(), packed_off.data(), index.data(),
                                        out_runs.data() + base, kept, &runs_home, (uint64_t)(base >> 2)};
                rc = hook(hook_user, chunk);
                if (rc) return rc;
            }
            if (!runs_home) MALCS_HIP(hipMemcpyAsync(out_runs.data() + base, d_packed.p, kept * 16ull, hipMemcpyDeviceToHost, st));   // pinned destination
            MALCS_HIP(malcs_sync(st));
            stages.lap("compaction, runs to the host");
        }
        {
            uint64_t last = base >> 2;                                        // a pair without runs ends where its predecessor ends
            uint32_t f = 0;
            for (uint32_t s = 0; s < all; s++) {
                if (f < cntf && keep[f] == s) { last = (base >> 2) + packed_off[f] + got[f]; f++; }
                out_first[pos + s + 1] = last;
            }
        }
        pos = end;
    }
    return MALCS_OK;
}

int malcs_sweep_small_dmax() { return SMALL_DMAX; }

// pairs per wave of the one-wave sweeps at distance D (the backward sweep's 2 D + 3 diagonals in a group's share of LDS):
// the caller orders its list by this number, so that a few distant pairs do not take the whole batch to one pair per wave
int malcs_sweep_pairs_per_wave(int64_t D)
{
    return ppw_of_diagonals(2 * D + 3);
}
