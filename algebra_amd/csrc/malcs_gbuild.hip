// malcs_gbuild.hip -- kernel family 2c: _build_graph on the device.
//
// Stands in for _build_graph(reference, observed, lcs_nodes, shift) (algebra/lcs/lcs_graph.py:351-453) for the pairs
// whose optimal runs the sweep kernels (malcs_sweep.hip) have just left in device memory: the runs are ordered as the
// reference's level lists (by level, sweep iteration, diagonal group, diagonal, row), the levels are
// linked by strict dominance of the run ends (:398), nodes are split (:405-412) and re-inserted (:418-423), source and
// sink are trimmed (:428-451), and the graph leaves in the numbering the C-ABI promises (BFS by identity from the
// source, edges in Node.edges order).  What comes back to the host per pair is the finished node / edge arrays; the
// host neither sorts nor links anything for these pairs.
//
// One wave per pair, two kernels.
//   * ordering (k_order_runs): lane-parallel.  Every lane computes the sort keys of its runs and their ranks by counting
//     the smaller keys (R^2 / 64 comparisons per lane against keys in LDS; R is a few dozen for a 1 kb window, at most
//     1 024 here) and puts them back in place: from then on the runs are final and travel home on the context's second
//     stream while the graphs are built;
//   * construction (k_graph_build): malcs_graph_core.hpp, the same source the host half compiles (so the two cannot drift apart), run
//     by lane 0 on an arena in LDS (graphs of <= 64 runs; everything is a few dependent LDS accesses per step) or in a
//     per-pair stretch of global scratch (up to 1 024 runs).  The construction is a chain of dependent list operations:
//     its parallelism is across the pairs of a batch (thousands of waves in flight), not inside a pair;
//   * output: lane 0 reserves the graph's words with one atomic add on a batch-wide cursor, so the host copies exactly
//     what was produced.  A small graph (LDS arena, out-degrees that add up to at most SE edges) is numbered and written
//     into LDS in one walk and copied out by the whole wave; a larger one is counted first and written by lane 0.
// A pair that does not fit (too many runs, arena overflow, output buffer full) is flagged and goes through the host
// half (malcs_build.cpp), which gives the same result.
#include "malcs_internal.hpp"
#include "malcs_graph_core.hpp"

namespace {

constexpr int PAIR_OUT = 10;            // per slot: status, n_nodes, n_edges, supremal[4], out offset (lo, hi), n_runs

// sort key of a run in the reference's emission order (malcs_build.cpp: order_runs)
__device__ __forceinline__ void run_key(const int4 r, int32_t delta, uint64_t &k1, uint64_t &k2)
{
    const int32_t re = r.x + r.z, ce = r.y + r.z, diag = r.y - r.x;
    const int32_t cost = re + ce - 2 * (r.w + 1);                        // forward edit cost of the corner after the run
    const int32_t off = diag - delta, aoff = off < 0 ? -off : off, adelta = delta < 0 ? -delta : delta;
    const int32_t it = (cost + aoff - adelta) / 2;                       // sweep iteration that first affords it
    const uint32_t grp = diag < delta ? 0u : (diag > delta ? 1u : 2u);
    const int32_t dk = grp == 0 ? diag : -diag;
    k1 = ((uint64_t)(uint32_t)r.w << 32) | ((uint64_t)(uint32_t)it << 2) | grp;
    k2 = ((uint64_t)(uint32_t)(dk + 0x40000000) << 32) | (uint32_t)r.x;
}

// The ordering alone, for every slot of a list, ahead of the construction: the runs are final after it (sorted in
// place), so that they travel home while the graphs are built (k_graph_build reads them as they lie).  Slots with no runs or more than RCAP are left alone (k_graph_build gives them up).
template <int RCAP>
__global__ void __launch_bounds__(64)
k_order_runs(uint4 *__restrict__ runs, const uint64_t *__restrict__ run_off, const uint32_t *__restrict__ run_cnt,
             const uint32_t *__restrict__ pids, const uint32_t *__restrict__ slots, uint32_t n_slots,
             const uint32_t *__restrict__ a_len, const uint32_t *__restrict__ b_len)
{
    __shared__ uint64_t key1[RCAP], key2[RCAP];
    if (blockIdx.x >= n_slots) return;
    const uint32_t slot = slots[blockIdx.x], lane = threadIdx.x;
    const uint32_t R = run_cnt[slot], pid = pids[slot];
    if (R < 2 || R > (uint32_t)RCAP) return;
    const int32_t delta = (int32_t)b_len[pid] - (int32_t)a_len[pid];
    uint4 *mine = runs + run_off[slot];
    constexpr int PER = (RCAP + 63) / 64;
    uint4 held[PER];
#pragma unroll
    for (int t = 0; t < PER; t++) {
        const uint32_t i = lane + 64u * t;
        if (i < R) {
            held[t] = mine[i];
            run_key(make_int4((int)held[t].x, (int)held[t].y, (int)held[t].z, (int)held[t].w), delta, key1[i], key2[i]);
        }
    }
    __syncthreads();                                                         // every run is in a register before any is overwritten
#pragma unroll
    for (int t = 0; t < PER; t++) {
        const uint32_t i = lane + 64u * t;
        if (i < R) {
            const uint64_t a1 = key1[i], a2 = key2[i];
            uint32_t rank = 0;
            for (uint32_t j = 0; j < R; j++) rank += (key1[j] < a1 || (key1[j] == a1 && key2[j] < a2)) ? 1u : 0u;
            mine[rank] = held[t];
        }
    }
}

// RCAP: runs a pair may have; NC / EC / LC: capacities of the LDS arena (LDS_ARENA) -- otherwise the arena is the slot's
// stretch of `scratch`, sized by the host from the slot's run count
template <int RCAP, bool LDS_ARENA, int NC, int EC, int LC>
__global__ void __launch_bounds__(64)
k_graph_build(const uint4 *__restrict__ runs, const uint64_t *__restrict__ run_off, const uint32_t *__restrict__ run_cnt,
              const uint32_t *__restrict__ pids, const uint32_t *__restrict__ slots, uint32_t n_slots,
              const uint32_t *__restrict__ a_len, const uint32_t *__restrict__ b_len, const int32_t *__restrict__ dist,
              const int64_t *__restrict__ shift, int32_t *__restrict__ scratch, const uint64_t *__restrict__ scratch_off,
              int32_t *__restrict__ out_words, unsigned long long out_cap, unsigned long long *__restrict__ out_cursor,
              int32_t *__restrict__ pair_out)
{
    // SE: edges of a graph that is written into LDS first (`stage`: nodes, then edges, int16) and copied out by the whole
    // wave; a graph with more goes the other way (counted, then written straight to memory by lane 0)
    constexpr int SE = LDS_ARENA ? (EC * 3) / 5 : 1;
    __shared__ int32_t sorted[4 * RCAP];
    __shared__ int32_t arena_mem[LDS_ARENA ? (9 * NC + 6 * EC + 2 * LC) : 1];
    __shared__ int16_t stage[LDS_ARENA ? 3 * NC + 6 * SE : 2];
    __shared__ int32_t bcast[4];
    __shared__ unsigned long long bcast_at;
    if (blockIdx.x >= n_slots) return;
    const uint32_t slot = slots[blockIdx.x], lane = threadIdx.x;
    const uint32_t R = run_cnt[slot], pid = pids[slot];
    int32_t *po = pair_out + (size_t)slot * PAIR_OUT;
    if (R == 0 || R > (uint32_t)RCAP) { if (lane == 0) { po[0] = 1; po[9] = (int32_t)R; } return; }
    const int32_t m = (int32_t)a_len[pid], n = (int32_t)b_len[pid];
    const uint4 *mine = runs + run_off[slot];
    // the runs lie in the reference's order (k_order_runs has been here)
    for (uint32_t i = lane; i < R; i += 64) {
        const uint4 u = mine[i];
        sorted[4 * i] = (int32_t)u.x; sorted[4 * i + 1] = (int32_t)u.y; sorted[4 * i + 2] = (int32_t)u.z; sorted[4 * i + 3] = (int32_t)u.w;
    }
    __syncthreads();
    malcs_core::arena A;                                                   // (every lane has the addresses, lane 0 the graph)
    if (LDS_ARENA) malcs_core::arena_bind(A, arena_mem, NC, EC, LC);
    else malcs_core::arena_bind(A, scratch + scratch_off[slot], 2 * (int32_t)R + 16, 8 * (int32_t)R + 64, (int32_t)R + 8);
    // the nodes of the runs, a lane per run (what add_node would do one by one in front of the construction)
    const bool filled = (int32_t)R <= A.node_cap;
    if (filled) {
        for (uint32_t r = lane; r < R; r += 64) {
            const int32_t len = sorted[4 * r + 2];
            A.row[r] = sorted[4 * r]; A.col[r] = sorted[4 * r + 1]; A.length[r] = len; A.wlen[r] = len;
            A.incoming[r] = 0; A.deg[r] = 0; A.tail[r] = -1;
        }
        __syncthreads();
    }
    int32_t source = 0, sink = 0, sup[4] = {0, 0, 0, 0}, status = 0, count = 0, R16 = 0;
    int64_t n_edges = 0;
    unsigned long long at = 0;
    if (lane == 0) {
        const int32_t n_levels = (m + n - dist[pid]) / 2;
        malcs_core::build(A, m, n, shift ? shift[pid] : 0, sorted, (int32_t)R, n_levels, source, sink, sup, malcs_core::no_growth(), filled);
        status = A.overflow ? 1 : 0;
        bcast[0] = status; bcast[1] = A.n_nodes;
    }
    bool staged = false;
    if constexpr (LDS_ARENA) {
        // Small graphs leave through LDS: the wave clears the numbering and adds up the out-degrees (an upper bound of the
        // edges the source reaches), lane 0 numbers and writes the graph into `stage` in one walk (no counting walk before
        // it), reserves its words, and the wave copies them out side by side.
        __syncthreads();
        const int32_t nn = bcast[0] ? 0 : bcast[1];
        int32_t degsum = 0;
        for (int32_t v = (int32_t)lane; v < nn; v += 64) { A.id[v] = -1; degsum += A.deg[v]; }
        for (int off = 32; off; off >>= 1) degsum += __shfl_xor(degsum, off, 64);
        staged = !bcast[0] && degsum <= SE && m < 32768 && n < 32768;
        __syncthreads();
    }
    if (lane == 0 && !status) {
        if (staged) {
            malcs_core::write_graph<int16_t>(A, source, stage, stage + 3 * NC, 0, nullptr, &count, &n_edges);
            const unsigned long long items = 3ull * (unsigned)count + 6ull * (unsigned long long)n_edges;
            const unsigned long long words = (items + 1ull) / 2ull;
            at = atomicAdd(out_cursor, words);
            if (at + words > out_cap) status = 1;                          // the output buffer is full: the host half takes the pair
            R16 = 1 << 30;
            bcast[2] = status ? -1 : count; bcast[3] = (int32_t)n_edges; bcast_at = at;
        } else {
            count = malcs_core::count_reachable(A, source, n_edges);
            // windows below 32 768 symbols: node triples and edge tuples as int16 (half the words to bring home -- 104 MB for
            // 100 000 C2 graphs otherwise); the cursor counts int32 words either way
            const bool half16 = m < 32768 && n < 32768 && count < 32768;
            const unsigned long long items = 3ull * (unsigned)count + 6ull * (unsigned long long)n_edges;
            const unsigned long long words = half16 ? (items + 1ull) / 2ull : items;
            if (n_edges > 0x7fffffffll) status = 1;
            else {
                at = atomicAdd(out_cursor, words);
                if (at + words > out_cap) status = 1;
                else if (half16) {
                    int16_t *o = reinterpret_cast<int16_t *>(out_words + at);
                    malcs_core::write_graph<int16_t>(A, source, o, o + 3ull * (unsigned)count, 0);
                } else malcs_core::write_graph<int32_t>(A, source, out_words + at, out_words + at + 3ull * (unsigned)count, 0);
                if (half16) R16 = 1 << 30;
            }
        }
    }
    if constexpr (LDS_ARENA) {
        __syncthreads();
        if (staged && bcast[2] >= 0) {
            const int32_t cnt = bcast[2], ne = bcast[3];
            int16_t *o = reinterpret_cast<int16_t *>(out_words + bcast_at);
            for (int32_t i = (int32_t)lane; i < 3 * cnt; i += 64) o[i] = stage[i];
            for (int32_t i = (int32_t)lane; i < 6 * ne; i += 64) o[3 * cnt + i] = stage[3 * NC + i];
        }
    }
    if (lane == 0) {
        po[0] = status; po[1] = count; po[2] = (int32_t)n_edges; po[3] = sup[0]; po[4] = sup[1]; po[5] = sup[2]; po[6] = sup[3];
        po[7] = (int32_t)(uint32_t)at; po[8] = (int32_t)(uint32_t)(at >> 32); po[9] = (int32_t)R | R16;        // bit 30: int16 words
    }
}

// ---- the bit-vector kernels' events -> runs ---------------------------------------------------------------------
// k_lcs_reach leaves, per pair, the cells where a run of optimal matches starts (kind 1) and ends (kind 0), in the order
// it met them.  Starts and ends alternate along every diagonal (malcs_build.cpp: runs_from_events): one wave per pair
// ranks its events by (diagonal, row, start before end) with the counting the graph kernel uses for its runs, pairs
// neighbours in that order into {row, col, length, level} and leaves the pair's runs where k_graph_build expects
// them.  Events that do not pair up give a count of 0: the pair stays with the host half, which reports the error.
template <int ECAP>
__global__ void __launch_bounds__(64)
k_events_to_runs(const uint4 *__restrict__ events, const uint64_t *__restrict__ ev_off, const uint32_t *__restrict__ ev_cnt,
                 const uint32_t *__restrict__ slots, uint32_t n_slots, uint4 *__restrict__ runs, const uint64_t *__restrict__ run_off,
                 uint32_t *__restrict__ run_cnt)
{
    __shared__ uint64_t key[ECAP];
    __shared__ uint4 sorted[ECAP];
    __shared__ uint32_t bad;
    if (blockIdx.x >= n_slots) return;
    const uint32_t slot = slots[blockIdx.x], lane = threadIdx.x;
    const uint32_t E = ev_cnt[slot];
    if (lane == 0) bad = (E & 1u) || E > (uint32_t)ECAP ? 1u : 0u;
    __syncthreads();
    if (bad) { if (lane == 0) run_cnt[slot] = 0u; return; }
    const uint4 *mine = events + ev_off[slot];
    for (uint32_t i = lane; i < E; i += 64) {
        const uint4 e = mine[i];                                         // row, col, level, kind
        const uint64_t diag = (uint64_t)(int64_t)((int64_t)e.y - (int64_t)e.x + 0x40000000ll);
        key[i] = (diag << 33) | ((uint64_t)e.x << 1) | (uint64_t)(1u - (e.w & 1u));
    }
    __syncthreads();
    for (uint32_t i = lane; i < E; i += 64) {
        const uint64_t a = key[i];
        uint32_t rank = 0;
        for (uint32_t j = 0; j < E; j++) rank += (key[j] < a || (key[j] == a && j < i)) ? 1u : 0u;
        sorted[rank] = mine[i];
    }
    __syncthreads();
    uint4 *out = runs + run_off[slot];
    for (uint32_t i = lane; i < E / 2u; i += 64) {
        const uint4 s = sorted[2 * i], e = sorted[2 * i + 1];
        const bool ok = s.w == 1u && e.w == 0u && (int64_t)s.y - (int64_t)s.x == (int64_t)e.y - (int64_t)e.x && e.x >= s.x;
        if (!ok) atomicOr(&bad, 1u);
        out[i] = make_uint4(s.x, s.y, e.x - s.x + 1u, e.z);
    }
    __syncthreads();
    if (lane == 0) run_cnt[slot] = bad ? 0u : E / 2u;
}

}  // namespace

int malcs_gbuild_pair_out_words() { return PAIR_OUT; }

// kind 0: graphs of at most 32 runs on a small arena in LDS (7 KB per wave with the sorted runs: twice as many waves per CU as kind 1, and
// the construction is a chain of dependent LDS accesses that only other waves can hide; 99 % of the graphs of 1 kb
// windows at 2 % indels); kind 1: at most 64 runs, 12.5 KB of LDS; kind 2: at most 1 024 runs on an arena in `scratch`
int malcs_gbuild_launch(malcs_ctx *ctx, int kind, uint4 *d_runs, const uint64_t *d_run_off, const uint32_t *d_run_cnt,
                        const uint32_t *d_pids, const uint32_t *d_slots, uint32_t n_slots, const uint32_t *d_a_len,
                        const uint32_t *d_b_len, const int32_t *d_dist, const int64_t *d_shift, int32_t *d_scratch,
                        const uint64_t *d_scratch_off, int32_t *d_out_words, uint64_t out_cap, unsigned long long *d_out_cursor,
                        int32_t *d_pair_out)
{
    if (!n_slots) return MALCS_OK;
    if (kind == 0)
        hipLaunchKernelGGL((k_graph_build<32, true, 64, 160, 40>), dim3(n_slots), dim3(64), 0, ctx->stream, d_runs, d_run_off, d_run_cnt,
                           d_pids, d_slots, n_slots, d_a_len, d_b_len, d_dist, d_shift, d_scratch, d_scratch_off, d_out_words,
                           (unsigned long long)out_cap, d_out_cursor, d_pair_out);
    else if (kind == 1)
        hipLaunchKernelGGL((k_graph_build<64, true, 112, 288, 72>), dim3(n_slots), dim3(64), 0, ctx->stream, d_runs, d_run_off, d_run_cnt,
                           d_pids, d_slots, n_slots, d_a_len, d_b_len, d_dist, d_shift, d_scratch, d_scratch_off, d_out_words,
                           (unsigned long long)out_cap, d_out_cursor, d_pair_out);
    else
        hipLaunchKernelGGL((k_graph_build<1024, false, 1, 1, 1>), dim3(n_slots), dim3(64), 0, ctx->stream, d_runs, d_run_off, d_run_cnt,
                           d_pids, d_slots, n_slots, d_a_len, d_b_len, d_dist, d_shift, d_scratch, d_scratch_off, d_out_words,
                           (unsigned long long)out_cap, d_out_cursor, d_pair_out);
    MALCS_HIP(hipGetLastError());
    return MALCS_OK;
}

// the runs of the slots of a list in the reference's order (kind as above: 0 and 1 at most 64 runs, 2 at most 1 024)
int malcs_gbuild_order(malcs_ctx *ctx, int kind, uint4 *d_runs, const uint64_t *d_run_off, const uint32_t *d_run_cnt, const uint32_t *d_pids,
                       const uint32_t *d_slots, uint32_t n_slots, const uint32_t *d_a_len, const uint32_t *d_b_len)
{
    if (!n_slots) return MALCS_OK;
    if (kind < 2) hipLaunchKernelGGL((k_order_runs<64>), dim3(n_slots), dim3(64), 0, ctx->stream, d_runs, d_run_off, d_run_cnt, d_pids, d_slots, n_slots, d_a_len, d_b_len);
    else hipLaunchKernelGGL((k_order_runs<1024>), dim3(n_slots), dim3(64), 0, ctx->stream, d_runs, d_run_off, d_run_cnt, d_pids, d_slots, n_slots, d_a_len, d_b_len);
    MALCS_HIP(hipGetLastError());
    return MALCS_OK;
}

// kind 0: at most 128 events per pair, kind 1: at most 2 048 (= the 1 024 runs the graph kernels take)
int malcs_gbuild_events_to_runs(malcs_ctx *ctx, int kind, const void *d_events, const uint64_t *d_ev_off, const uint32_t *d_ev_cnt,
                                const uint32_t *d_slots, uint32_t n_slots, uint4 *d_runs, const uint64_t *d_run_off, uint32_t *d_run_cnt)
{
    if (!n_slots) return MALCS_OK;
    const uint4 *ev = static_cast<const uint4 *>(d_events);
    if (kind == 0) hipLaunchKernelGGL((k_events_to_runs<128>), dim3(n_slots), dim3(64), 0, ctx->stream, ev, d_ev_off, d_ev_cnt, d_slots, n_slots, d_runs, d_run_off, d_run_cnt);
    else hipLaunchKernelGGL((k_events_to_runs<2048>), dim3(n_slots), dim3(64), 0, ctx->stream, ev, d_ev_off, d_ev_cnt, d_slots, n_slots, d_runs, d_run_off, d_run_cnt);
    MALCS_HIP(hipGetLastError());
    return MALCS_OK;
}
