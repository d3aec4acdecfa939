// malcs_internal.hpp -- shared host-side declarations of libmalcs (not part of the C-ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <string>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <cstring>
#include <new>
#include "../../include/malcs.h"

void malcs_set_error(const char *fmt, ...);

#define MALCS_HIP(call)                                                                       \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess) {                                                               \
            malcs_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            return MALCS_ERR_HIP;                                                             \
        }                                                                                     \
    } while (0)

// MALCS_TRACE=1: wall time of the stages of a host-side call, to stderr (dev aid; costs one getenv per call when off)
struct malcs_stage_trace {
    bool on;
    const char *what;
    std::chrono::steady_clock::time_point t0;
    explicit malcs_stage_trace(const char *w) : on(getenv("MALCS_TRACE") != nullptr), what(w), t0(std::chrono::steady_clock::now()) {}
    void lap(const char *stage)
    {
        if (!on) return;
        const auto t1 = std::chrono::steady_clock::now();
        fprintf(stderr, "[malcs] %s: %s %.2f ms\n", what, stage, std::chrono::duration<double, std::milli>(t1 - t0).count());
        t0 = t1;
    }
};

// device memory that is released on every return path of the function that holds it
template <typename T> struct dev_scoped {
    T *p = nullptr;
    dev_scoped() = default;
    dev_scoped(const dev_scoped &) = delete;
    dev_scoped &operator=(const dev_scoped &) = delete;
    ~dev_scoped() { if (p) (void)hipFree(p); }
};


// A device-to-host copy that went through the staging buffer: lands in `dst` at the next malcs_sync.
struct staged_copy { void *dst; const void *src; size_t bytes; };

struct malcs_ctx {
    int device = -1;
    hipStream_t stream = nullptr;
    hipStream_t copy_stream = nullptr;     // made when first needed: results that travel home while the next kernels run
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr;
    hipDeviceProp_t prop;
    // pinned staging for the many small parameter / result copies of a call (a copy from or to pageable memory
    // costs ~16 us of host time each, from pinned memory it is an enqueue)
    char *stage = nullptr;
    size_t stage_cap = 0, stage_used = 0;
    std::vector<staged_copy> staged;
    uint32_t *noalloc_scratch = nullptr;   // this device's copy of the file-scope scratch array of malcs_noalloc.hip
    // Device blocks that batches of this context have released, kept for the next batch: a hipMalloc costs a fraction
    // of a millisecond and hipFree synchronises the device, which is most of what a small batch would otherwise pay.
    struct pool_block { void *p; size_t bytes; bool used; };
    std::vector<pool_block> pool;
    size_t pool_bytes = 0;
    // pinned host buffer for input that is packed on the host (malcs_batch_create): grows to the largest batch seen
    char *pack_stage = nullptr;
    size_t pack_stage_cap = 0;
    // host arrays of batch creation that are only needed while it runs, and batch objects that were destroyed: kept so
    // that the next batch does not pay for fresh pages (a 100 000-pair batch touches 6 MB of them)
    std::vector<uint64_t> seq_byte_off;
    std::vector<uint32_t> seq_len, seq_blk_off;
    std::vector<struct malcs_batch *> spare_batches;
    // host threads for the parallel stretches of batch creation (malcs_hostpack.hpp), started with the first large batch
    class malcs_workers *workers = nullptr;
};
class malcs_workers *malcs_ctx_workers(malcs_ctx *ctx);   // at most 16 (MALCS_PACK_THREADS), never more than the machine has

// Pinned host memory from a process-wide pool (blocks are kept when they are released: pinning a hundred megabytes
// costs tens of milliseconds, and results that come home into pageable memory pay for every fresh page).
void *malcs_pinned_alloc(size_t bytes);        // nullptr when the runtime has none to give
void malcs_pinned_free(void *p);

// a growable array in pinned memory (what device results are copied into and then read from in place)
template <typename T> struct pinned_vec {
    T *p = nullptr;
    size_t n = 0, cap = 0;
    pinned_vec() = default;
    pinned_vec(const pinned_vec &) = delete;
    pinned_vec &operator=(const pinned_vec &) = delete;
    pinned_vec(pinned_vec &&o) noexcept : p(o.p), n(o.n), cap(o.cap) { o.p = nullptr; o.n = o.cap = 0; }
    ~pinned_vec() { if (p) malcs_pinned_free(p); }
    size_t size() const { return n; }
    bool empty() const { return n == 0; }
    T *data() { return p; }
    const T *data() const { return p; }
    T &operator[](size_t i) { return p[i]; }
    const T &operator[](size_t i) const { return p[i]; }
    void clear() { n = 0; }
    void resize(size_t m)                          // new elements are NOT initialised
    {
        if (m > cap) {
            const size_t want = std::max(m, cap + cap / 2);
            T *q = static_cast<T *>(malcs_pinned_alloc(std::max<size_t>(want, 1) * sizeof(T)));
            if (!q) throw std::bad_alloc();
            if (n) memcpy(q, p, n * sizeof(T));
            if (p) malcs_pinned_free(p);
            p = q; cap = want;
        }
        n = m;
    }
};

// Pooled device memory of a context (the context's device must be current).  malcs_pool_free never synchronises: the
// block may still be read by work queued on the context's stream, and is only handed out again to work queued later
// on that same stream.
hipError_t malcs_pool_alloc(malcs_ctx *c, void **p, size_t bytes);
void malcs_pool_free(malcs_ctx *c, void *p);
void malcs_pool_trim(malcs_ctx *c);            // hipFree everything that is not in use

// pooled device memory that is released on every return path of the function that holds it
template <typename T> struct pool_scoped {
    malcs_ctx *c;
    T *p = nullptr;
    explicit pool_scoped(malcs_ctx *ctx) : c(ctx) {}
    pool_scoped(const pool_scoped &) = delete;
    pool_scoped &operator=(const pool_scoped &) = delete;
    hipError_t alloc(size_t count) { return malcs_pool_alloc(c, (void **)&p, count * sizeof(T)); }
    ~pool_scoped() { if (p) malcs_pool_free(c, p); }
};

// Copies on a context's stream and the wait for them.  Small copies are staged through the context's pinned buffer:
// host-to-device data is captured at the call (the source may be reused at once), device-to-host data reaches its
// destination when malcs_sync returns -- so results are read after malcs_sync, never after another kind of wait.
hipError_t malcs_h2d(hipStream_t st, void *dst, const void *src, size_t bytes);
hipError_t malcs_d2h(hipStream_t st, void *dst, const void *src, size_t bytes);
hipError_t malcs_sync(hipStream_t st);

// One launch of a family-1 kernel: a class of pairs that share (words per lane, lanes per pair).
struct dist_bucket {
    int W = 0;              // 64-bit words per lane
    int L = 0;              // lanes per pair (power of two)
    uint32_t first = 0;     // range [first, first + count) of batch->order
    uint32_t count = 0;
};

// Window kernel (malcs_diag.hip): similar pairs, one lane per pair, a few limbs around the main diagonal.
struct diag_retry {        // pairs the window could not settle go back to their full-matrix class
    int W = 0, L = 0;
    uint32_t first = 0;    // range [first, first + cap) of d_retry_list
    uint32_t cap = 0;
};
struct diag_plan {
    int nlw = 0;                         // limbs in the window (0: no window class in this batch)
    int T = 0;                           // results <= T are exact (the threshold of three-limb tiles)
    uint32_t n_tiles = 0, n_pairs = 0;   // of the class (from the lengths alone)
    uint32_t n_launch = 0;               // workgroups of a launch: an upper bound of the pieces (those beyond the plan's count leave at once)
    uint32_t max_len = 0, HB = 0;        // longest sequence of the class; bins per group of the block-count histogram
    bool can_half = false, cut_possible = false, sample_inside = false;   // sample_inside: the sample runs in the plan's first launch
    uint32_t n_sample = 0;               // sampled pairs (two sub-pairs each)
    int sample_W = 0, sample_L = 0;
    uint64_t rows_cap = 0;               // rows (of 64 dwords) the row buffer holds
    // One allocation (d_rows is its start); everything below is a part of it.  Static = written once per batch (from
    // the lengths); the rest is rewritten by every plan, on the device.
    uint32_t *d_rows = nullptr;          // per piece, lane-interleaved rows of 64 dwords
    uint32_t *d_pairs = nullptr;         // static: pair ids of the class
    uint8_t *d_cls = nullptr;            // static: their retry class
    uint32_t *d_item = nullptr, *d_sitem = nullptr, *d_pub = nullptr, *d_sets = nullptr;   // item records by pair / by slot; the published plan;   // sets: two of { histogram, cursors, column sums }, used by alternate plans
    uint32_t set_words = 0, plan_parity = 0;
    uint32_t *d_lay_off = nullptr, *d_lay_info = nullptr, *d_lay_it = nullptr;   // the pieces in the order of the tiles: first row, column blocks | mode flags (0 = no piece)
    uint32_t *d_piece_off = nullptr;     // the same in the order of the launch's workgroups: first row of each piece
    uint32_t *d_piece_info = nullptr;    // column blocks of each piece | mode flags; 0 = no piece
    uint32_t *d_piece_join = nullptr;    // for the pieces of a cut tile: which slot of the join buffer they meet in
    uint32_t *d_piece_place = nullptr;   // which place of the layout a workgroup runs (for whoever wants to look at the plan)
    uint32_t *d_join_buf = nullptr;
    uint32_t *d_stats = nullptr;         // what the last plan decided (plan_stats)
    uint32_t *d_sample = nullptr;        // static: a_off, a_len, b_off, b_len, list of the sample's sub-pairs, a correction per sampled pair, the count
    uint32_t *d_sample_tab = nullptr;    // static: the sampled pairs {a_off, a_len, b_off, b_len} (the one-lane sample cuts them itself)
    bool sample_lanes = false;           // the sample runs one lane per part of a pair (llcs_band32_lane) ...
    int sample_log_parts = 0;            // ... of which a pair has 1 << this many
    int32_t *d_sample_dist = nullptr;    // the sub-pairs' distances (written by every plan)
    uint32_t *d_retry_list = nullptr, *d_retry_first = nullptr, *d_retry_cap = nullptr, *d_retry_count = nullptr;
    std::vector<diag_retry> retry;
    uint32_t parity = 0;                 // which of the two sets of retry counters the last launch used
    bool planned = false;                // a plan has been queued
};

// what a plan decided, in device memory (dwords; read back only when somebody asks: malcs_diag_stats)
enum { PS_NPIECES, PS_NHALF, PS_NCUT, PS_HALFMODE, PS_ERROR, PS_ROWS, PS_BLOCKS, PS_BLOCKS_HALF, PS_IT_LAST, PS_LAST_HALF,
       PS_COLS0_LO, PS_COLS0_HI, PS_COLS1_LO, PS_COLS1_HI, PS_SAMPLE_MEAN_X16, PS_SAMPLE_SD_X16, PS_WORST,
       PS_BALANCED, PS_MINOR_A, PS_NAB, PS_FIELDS };

struct malcs_batch {
    malcs_ctx *ctx = nullptr;
    uint64_t n_pairs = 0;
    int n_planes = 2;
    // device
    uint32_t *d_planes = nullptr;        // packed arena: per sequence, per 32-symbol block, n_planes dwords
    uint32_t *d_a_off = nullptr, *d_a_len = nullptr, *d_b_off = nullptr, *d_b_len = nullptr;   // per pair (dword offsets)
    uint32_t *d_order = nullptr;         // pair ids grouped by bucket
    int32_t *d_dist = nullptr;
    uint32_t *d_pair_block = nullptr;    // the allocation d_a_off ... d_dist are parts of
    // host mirrors
    std::vector<uint32_t> a_off, a_len, b_off, b_len;   // dword offsets into d_planes / lengths in symbols
    std::vector<uint8_t> plan_cls, plan_in_window, plan_diag_cls;   // scratch of malcs_dist_plan, kept with the object (pages)
    std::vector<uint32_t> plan_diag_pairs;
    uint32_t plan_window_longest = 0;                               // longest sequence among the pairs of the window class
    uint64_t plan_window_blocks = 0;                                // sum of their texts' 32-column blocks (sizes the row buffer)
    std::vector<dist_bucket> dist_buckets;
    std::vector<uint32_t> band_pairs;    // pairs too long for a resident pattern: sliding-window kernel
    diag_plan diag;
    uint64_t full_cells = 0;             // sum of m * n over the batch
    uint64_t arena_dwords = 0;
    uint32_t max_len = 0;
};

// malcs_api.cpp: a batch whose symbols (one byte each) already sit in device memory; offsets and lengths on the host
int malcs_batch_create_device(malcs_ctx *ctx, const uint8_t *d_syms, uint64_t n_syms,
                              const uint64_t *a_off, const uint32_t *a_len, const uint64_t *b_off, const uint32_t *b_len,
                              uint64_t n_pairs, int n_symbols, malcs_batch **out);

// malcs_noalloc.hip
int malcs_noalloc_setup(malcs_ctx *c);

// malcs_pack.hip
int malcs_pack_launch(malcs_ctx *ctx, const uint8_t *d_syms, const uint64_t *d_seq_byte_off,
                      const uint32_t *d_seq_len, const uint32_t *d_seq_blk_off, uint32_t n_seqs,
                      uint32_t total_blocks, int n_planes, uint32_t *d_planes);
int malcs_clear_tails_launch(malcs_ctx *ctx, uint32_t *d_planes, const uint32_t *d_a_off, const uint32_t *d_a_len,
                             const uint32_t *d_b_off, const uint32_t *d_b_len, uint32_t n_pairs, int n_planes);

// malcs_diag.hip
// pairs[i] goes to the window class; retry_class[i] indexes `classes` (its full-matrix launch class)
// setup: once per batch, from the lengths alone (allocation, tables); plan: everything that depends on what the
// sequences hold (common ends, sort, tiles, pieces, rows), queued on the context's stream, nothing comes back
int malcs_diag_setup(malcs_batch *b, int nlw, const std::vector<uint32_t> &pairs, const std::vector<uint8_t> &retry_class,
                     const std::vector<diag_retry> &classes);
int malcs_diag_plan(malcs_batch *b);
int malcs_diag_stats(malcs_batch *b, uint32_t *stats);
int malcs_diag_dump(malcs_batch *b, int what, uint32_t *out, uint64_t cap, uint64_t *n_out);                 // synchronises; PS_FIELDS dwords
int malcs_diag_launch(malcs_batch *b, bool advance, hipEvent_t start = nullptr, hipEvent_t stop = nullptr);   // events: the launch's own start and end
void malcs_diag_release(malcs_batch *b);
int malcs_diag_threshold(int nlw);

// malcs_dist.hip
int malcs_dist_plan(malcs_batch *b);          // classes from the lengths (host) + malcs_dist_replan
int malcs_dist_replan(malcs_batch *b);        // what depends on the sequences: the window class's plan, on the device
int malcs_dist_launch(malcs_batch *b, hipEvent_t window_start = nullptr, hipEvent_t window_stop = nullptr);   // events: around the window kernel of the step
int malcs_dist_launch_resident(malcs_batch *b, hipEvent_t window_start = nullptr, hipEvent_t window_stop = nullptr);
int malcs_dist_band(malcs_batch *b, const std::vector<uint32_t> &pairs, int32_t *host_out);
int malcs_dist_launch_list(hipStream_t st, int n_planes, int W, int L, const uint32_t *d_planes,
                           const uint32_t *d_a_off, const uint32_t *d_a_len, const uint32_t *d_b_off, const uint32_t *d_b_len,
                           const uint32_t *d_list, uint32_t first, const uint32_t *d_count, uint32_t cap, uint32_t waves,
                           int32_t *d_dist);

// malcs_sweep.hip
// A chunk of pairs whose optimal runs have just been compacted in device memory (before they are copied to the host):
// slot f has runs d_runs[run_off[f] .. + cnt[f]) = {row, col, length, level}, belongs to pair pids[f] and to entry
// work_index[f] of the caller's work list.  host_runs (pinned, n_runs * 4 words; may be null): where the caller wants
// the chunk's runs -- a hook that leaves them final before it is done may send them there itself, on a stream of its
// own, and says so in *runs_home (the caller then does not copy them again).
struct malcs_sweep_chunk {
    uint4 *d_runs; const uint64_t *d_run_off; const uint32_t *d_run_cnt; const uint32_t *d_pids;
    uint32_t count;
    const uint32_t *pids, *cnt; const uint64_t *run_off, *work_index;      // host copies
    uint32_t *host_runs; uint64_t n_runs; bool *runs_home;
    uint64_t runs_base;                                                    // runs of earlier chunks in the caller's array (host_runs begins behind them)
};
typedef int (*malcs_sweep_hook)(void *user, const malcs_sweep_chunk &chunk);
int malcs_sweep_distances(malcs_batch *b, const std::vector<uint32_t> &pairs, int32_t cap, std::vector<int32_t> &out);

// malcs_comm.cpp
int malcs_comm_all_gather_async(malcs_comm *c, const void *d_send, void *d_recv, uint64_t bytes);
int malcs_comm_world(const malcs_comm *c);
int malcs_comm_rank(const malcs_comm *c);
malcs_ctx *malcs_comm_ctx(const malcs_comm *c);
