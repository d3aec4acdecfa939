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

template <typename T> struct dbuf {
    T *p = nullptr;
    ~dbuf() { if (p) hipFree(p); }
    int alloc(size_t n) { return hipMalloc(&p, (n ? n : 1) * sizeof(T)) == hipSuccess ? 0 : -1; }
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
    size_t free_b = 0, total_b = 0;
    MALCS_HIP(hipMemGetInfo(&free_b, &total_b));
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
        dbuf<uint32_t> d_work, d_counts;
        dbuf<int32_t> d_fr, d_reached;
        dbuf<uint64_t> d_fr_off, d_out_off;
        dbuf<uint4> d_out;
        if (d_work.alloc(cnt) || d_counts.alloc(cnt) || d_fr.alloc(fr_total) || d_reached.alloc(cnt) || d_fr_off.alloc(cnt) ||
            d_out_off.alloc(cnt)) {
            malcs_set_error("device allocation failed (sweep chunk of %u pairs)", cnt);
            return MALCS_ERR_NOMEM;
        }
        MALCS_HIP(hipMemcpyAsync(d_work.p, work.data() + pos, cnt * 4ull, hipMemcpyHostToDevice, st));
        MALCS_HIP(hipMemcpyAsync(d_fr_off.p, fr_off.data(), cnt * 8ull, hipMemcpyHostToDevice, st));
        MALCS_HIP(hipMemsetAsync(d_reached.p, 0xff, cnt * 4ull, st));
        int rc = launch_any(b, false, cnt, d_work.p, d_fr.p, d_fr_off.p, d_counts.p, nullptr, nullptr, d_reached.p);
        if (rc) return rc;
        std::vector<uint32_t> counts(cnt);
        std::vector<int32_t> reached(cnt);
        MALCS_HIP(hipMemcpyAsync(counts.data(), d_counts.p, cnt * 4ull, hipMemcpyDeviceToHost, st));
        MALCS_HIP(hipMemcpyAsync(reached.data(), d_reached.p, cnt * 4ull, hipMemcpyDeviceToHost, st));
        MALCS_HIP(hipStreamSynchronize(st));
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
        MALCS_HIP(hipMemcpyAsync(d_out_off.p, off.data(), cnt * 8ull, hipMemcpyHostToDevice, st));
        rc = launch_any(b, true, cnt, d_work.p, d_fr.p, d_fr_off.p, d_counts.p, d_out.p, d_out_off.p, d_reached.p);
        if (rc) return rc;
        const size_t base = out_runs.size();
        out_runs.resize(base + total * 4);
        if (total) MALCS_HIP(hipMemcpyAsync(out_runs.data() + base, d_out.p, total * 16ull, hipMemcpyDeviceToHost, st));
        MALCS_HIP(hipStreamSynchronize(st));
        for (uint32_t s = 0; s < cnt; s++) out_first[pos + s + 1] = base / 4 + off[s] + counts[s];
        pos = end;
    }
    return MALCS_OK;
}
