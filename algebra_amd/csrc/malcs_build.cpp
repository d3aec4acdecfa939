// malcs_build.cpp -- host half of family 2: drives the two device passes, turns the device's
// run-boundary events into level-ordered runs and compresses them into the LCS graph.
//
// Reference behaviour restated here (paths relative to the reference checkout):
//   * ordering of runs inside a level = order in which the O(NP) sweep of _lcs_nodes meets them
//     (algebra/lcs/lcs_graph.py:335-342: diagonals below delta ascending, above delta descending,
//     delta last, iteration by iteration; within a diagonal by position) -- SURVEY 8(a) P4;
//   * _build_graph (algebra/lcs/lcs_graph.py:351-453): degenerate cases :370-377, sink :379-385,
//     level loop with edge creation / node splitting / re-insertion :389-426, source :428-445,
//     trimming of source and sink :447-451, supremal :453.
// The graph builder works on integer node ids and flat arrays; nothing here touches symbols.
#include "malcs_internal.hpp"
#include "malcs_hostpack.hpp"
#include "malcs_graph_core.hpp"
#include <algorithm>
#include <atomic>
#include <chrono>
#include <deque>
#include <memory>
#include <mutex>
#include <cstring>
#include <new>
#include <thread>

int malcs_graph_max_band_words(int n_planes);
int malcs_graph_launch_forward(malcs_batch *b, int W, int logL, bool band, const uint32_t *d_work, uint32_t count,
                               uint32_t *d_scratch, const uint64_t *d_sc_off, const int32_t *d_kmin,
                               const int32_t *d_kmax, const int32_t *d_rw, int32_t *d_dist_out);
int malcs_graph_launch_reach(malcs_batch *b, int W, int logL, bool band, const uint32_t *d_work, uint32_t count,
                             const uint32_t *d_scratch, const uint64_t *d_sc_off, const int32_t *d_kmin,
                             const int32_t *d_kmax, const int32_t *d_rw, const int32_t *d_llcs, const uint8_t *d_skip,
                             void *d_events, const uint64_t *d_ev_off, const uint32_t *d_ev_cap, uint32_t *d_ev_cnt);
int malcs_graph_launch_compact(malcs_ctx *ctx, const void *d_events, const uint64_t *d_ev_off, const uint32_t *d_ev_n,
                               const uint64_t *d_out_off, uint32_t count, void *d_out);

int malcs_sweep_runs(malcs_batch *b, const std::vector<uint32_t> &work, const int32_t *dist,
                     std::vector<uint32_t> &out_runs, std::vector<uint64_t> &out_first);

int malcs_sweep_optimal_runs(malcs_batch *b, bool small, const std::vector<uint32_t> &work, int32_t *dist, int32_t unknown_cap,
                             pinned_vec<uint32_t> &out_runs, std::vector<uint64_t> &out_first, float *ms,
                             malcs_sweep_hook hook = nullptr, void *hook_user = nullptr);
int malcs_gbuild_pair_out_words();
int malcs_gbuild_launch(malcs_ctx *ctx, int kind, uint4 *d_runs, const uint64_t *d_run_off, const uint32_t *d_run_cnt,
                        const uint32_t *d_pids, const uint32_t *d_slots, uint32_t n_slots, const uint32_t *d_a_len,
                        const uint32_t *d_b_len, const int32_t *d_dist, const int64_t *d_shift, int32_t *d_scratch,
                        const uint64_t *d_scratch_off, int32_t *d_out_words, uint64_t out_cap, unsigned long long *d_out_cursor,
                        int32_t *d_pair_out);
int malcs_gbuild_order(malcs_ctx *ctx, int kind, uint4 *d_runs, const uint64_t *d_run_off, const uint32_t *d_run_cnt, const uint32_t *d_pids,
                       const uint32_t *d_slots, uint32_t n_slots, const uint32_t *d_a_len, const uint32_t *d_b_len);
int malcs_sweep_small_dmax();
int malcs_sweep_pairs_per_wave(int64_t D);
int malcs_gbuild_events_to_runs(malcs_ctx *ctx, int kind, const void *d_events, const uint64_t *d_ev_off, const uint32_t *d_ev_cnt,
                                const uint32_t *d_slots, uint32_t n_slots, uint4 *d_runs, const uint64_t *d_run_off, uint32_t *d_run_cnt);

namespace {

struct event { uint32_t row, col, level, kind; };      // kind 0 = run ends in (row, col); 1 = run starts in (row, col)

struct run { int64_t row, col, len, level; };

// The result arrays of all pairs a worker thread built sit back to back in that thread's blocks (no allocation per
// pair); a pair holds views.
struct i64_view {
    int64_t *p = nullptr;
    size_t n = 0;
    size_t size() const { return n; }
    bool empty() const { return n == 0; }
    int64_t *data() const { return p; }
    int64_t &operator[](size_t i) const { return p[i]; }
};

struct i64_store {
    std::deque<std::vector<int64_t>> blocks;
    size_t used = 0;
    i64_view take(size_t count)
    {
        if (blocks.empty() || used + count > blocks.back().size()) {
            blocks.emplace_back(std::max<size_t>(count, (size_t)1 << 18));
            used = 0;
        }
        i64_view v{blocks.back().data() + used, count};
        used += count;
        return v;
    }
};

// The int32 arrays of the graphs a worker thread built, back to back in blocks of raw memory (nothing is cleared: every
// word is written by write_graph).  Blocks of the default size go back to a process-wide free list when the result
// handle goes and are taken from there first: a batch of tandem-repeat windows writes a hundred megabytes of edges,
// and fresh pages for them (first touch, sixteen threads at once) cost more than writing them.
constexpr size_t I32_BLOCK = (size_t)1 << 22;             // words: 16 MB
struct i32_blocks {
    std::mutex lock;
    std::vector<int32_t *> free_list;                      // I32_BLOCK words each, at most 32 (512 MB)
    int32_t *get() { std::lock_guard<std::mutex> g(lock); if (free_list.empty()) return nullptr; int32_t *p = free_list.back(); free_list.pop_back(); return p; }
    bool put(int32_t *p) { std::lock_guard<std::mutex> g(lock); if (free_list.size() >= 32) return false; free_list.push_back(p); return true; }
    ~i32_blocks() { for (int32_t *p : free_list) delete[] p; }
};
i32_blocks &spare_i32_blocks() { static i32_blocks b; return b; }

struct i32_store {
    struct block { int32_t *p; size_t words; };
    std::vector<block> blocks;
    size_t used = 0;
    i32_store() = default;
    i32_store(const i32_store &) = delete;
    i32_store &operator=(const i32_store &) = delete;
    int32_t *take(size_t count)
    {
        if (blocks.empty() || used + count > blocks.back().words) {
            int32_t *p = count <= I32_BLOCK ? spare_i32_blocks().get() : nullptr;
            const size_t words = std::max(count, I32_BLOCK);
            if (!p) p = new int32_t[words];
            blocks.push_back({p, words});
            used = 0;
        }
        int32_t *at = blocks.back().p + used;
        used += count;
        return at;
    }
    ~i32_store() { for (block &b : blocks) if (b.words != I32_BLOCK || !spare_i32_blocks().put(b.p)) delete[] b.p; }
};

struct pair_graph {
    // first what every pass over the handles of a batch touches (112 bytes: a batch of 100 000 is a 20 MB array, and the
    // passes -- launch classes, the device's graphs pointed at, the columns a caller reads -- cost what they miss in cache)
    int32_t status = MALCS_PAIR_OK;
    // narrow: the graph's arrays as int32 / int16 words, coordinates without the shift (what malcs_graph_core.hpp writes,
    // on the host or on the device: dev_built, malcs_gbuild.hip); `nodes` / `edges` below are filled from them when somebody
    // needs the int64 arrays in place (widen()), the copy-out entry points widen on the way
    bool narrow = false, dev_built = false, canon_done = false;
    uint8_t dev_runs_in = 0;        // 0: dev_runs is an address; k + 1: an index into the flat array of sweep set k (see below)
    int64_t distance = 0;
    int64_t n_levels = 0;
    int64_t supremal[4] = {0, 0, 0, 0};
    int64_t dev_n_nodes = 0, dev_n_edges = 0, dev_n_runs = 0, dev_shift = 0;
    const int16_t *dev_nodes16 = nullptr, *dev_edges16 = nullptr;      // the words as int16 (graphs the device built over windows below 32 768)
    // its runs as the kernels left them (sorted, without the shift), 4 per run: where they are, or (dev_runs_in > 0) the
    // index of the first in the flat array of that sweep set, which may still grow while later chunks come home.
    // (Every member starts as zero: a batch's array of handles is made by clearing it, 0.3 ms per 100 000; one member
    // that started as -1 made that 1.6 ms.)
    union { const uint32_t *dev_runs = nullptr; uint64_t dev_runs_at; };
    // ... then what a reader of one graph needs
    const int32_t *dev_nodes = nullptr, *dev_edges = nullptr;          // the words as int32
    i64_view runs;                  // 4 per run (shifted row, col; length; level)
    i64_view nodes;                 // 3 per node
    i64_view edges;                 // 6 per edge
    std::vector<int64_t> canon;     // 4 per canonical variant (start, end, obs_from, obs_to), filled on demand
    int64_t node_word(size_t i) const { return dev_nodes16 ? dev_nodes16[i] : dev_nodes[i]; }
    int64_t edge_word(size_t i) const { return dev_edges16 ? dev_edges16[i] : dev_edges[i]; }
    size_t n_nodes() const { return narrow ? (size_t)dev_n_nodes : nodes.size() / 3; }
    size_t n_edges() const { return narrow ? (size_t)dev_n_edges : edges.size() / 6; }
    size_t n_runs() const { return dev_built ? (size_t)dev_n_runs : runs.size() / 4; }
};

// ---- events -> runs in the reference's emission order -------------------------------------------
// a pair's events: a slice of the chunk buffer they came home in (sorted in place)
struct event_span {
    event *p = nullptr;
    size_t n = 0;
    event *begin() const { return p; }
    event *end() const { return p + n; }
    size_t size() const { return n; }
    const event &operator[](size_t i) const { return p[i]; }
};

// the reference's emission order inside a level: by sweep iteration, then diagonals below
// delta ascending, above delta descending, delta itself last, then along the diagonal
void order_runs(std::vector<run> &out, int64_t m, int64_t n)
{
    const int64_t delta = n - m, adelta = delta < 0 ? -delta : delta;
    struct keyed { int64_t level, it, grp, dk, row; run r; };
    static thread_local std::vector<keyed> k;             // reused by the worker thread from pair to pair
    k.clear();
    k.reserve(out.size());
    for (const run &r : out) {
        const int64_t re = r.row + r.len, ce = r.col + r.len, diag = r.col - r.row;
        const int64_t cost = re + ce - 2 * (r.level + 1);                    // forward edit cost of the corner after the run
        const int64_t off = diag - delta, aoff = off < 0 ? -off : off;
        const int64_t it = (cost + aoff - adelta) / 2;                       // sweep iteration that first affords it
        const int64_t grp = diag < delta ? 0 : (diag > delta ? 1 : 2);
        k.push_back({r.level, it, grp, grp == 0 ? diag : -diag, r.row, r});
    }
    std::sort(k.begin(), k.end(), [](const keyed &a, const keyed &b) {
        if (a.level != b.level) return a.level < b.level;
        if (a.it != b.it) return a.it < b.it;
        if (a.grp != b.grp) return a.grp < b.grp;
        if (a.dk != b.dk) return a.dk < b.dk;
        return a.row < b.row;
    });
    for (size_t i = 0; i < k.size(); i++) out[i] = k[i].r;
}

bool runs_from_events(event_span ev, int64_t m, int64_t n, std::vector<run> &out)
{
    // starts and ends alternate along every diagonal
    std::sort(ev.begin(), ev.end(), [](const event &a, const event &b) {
        const int64_t da = (int64_t)a.col - a.row, db = (int64_t)b.col - b.row;
        if (da != db) return da < db;
        if (a.row != b.row) return a.row < b.row;
        return a.kind > b.kind;                      // a start sorts before the end of the same cell
    });
    if (ev.size() % 2) return false;
    out.clear();
    out.reserve(ev.size() / 2);
    for (size_t i = 0; i < ev.size(); i += 2) {
        const event &s = ev[i], &e = ev[i + 1];
        if (s.kind != 1 || e.kind != 0) return false;
        if ((int64_t)s.col - s.row != (int64_t)e.col - e.row || e.row < s.row) return false;
        out.push_back({s.row, s.col, (int64_t)e.row - s.row + 1, e.level});
    }
    order_runs(out, m, n);
    return true;
}

// ---- _build_graph -------------------------------------------------------------------------------
// The construction itself is malcs_graph_core.hpp (one source for this host path and for the device kernel of
// malcs_gbuild.hip).  A worker thread keeps one arena and reuses it from pair to pair: no allocation per pair or node.
struct builder {
    std::vector<int32_t> node[9], edge, list[2], runs32, where;      // the arena's arrays, each its own vector
    malcs_core::arena A;
    void bind()
    {
        int32_t **n[9] = {&A.row, &A.col, &A.length, &A.wlen, &A.incoming, &A.deg, &A.tail, &A.id, &A.order};
        for (int k = 0; k < 9; k++) *n[k] = node[k].data();
        A.edge = edge.data();
        A.node_cap = (int32_t)node[0].size(); A.edge_cap = (int32_t)(edge.size() / 6); A.list_cap = (int32_t)list[0].size();
    }
    void start(int32_t nodes, int32_t edges, int32_t lists)
    {
        for (auto &v : node) if ((int32_t)v.size() < nodes) v.resize((size_t)nodes);
        if (edge.size() < 6 * (size_t)edges) edge.resize(6 * (size_t)edges);
        for (auto &v : list) if ((int32_t)v.size() < lists) v.resize((size_t)lists);
        // arrays grown for one graph stay for the next, all of a kind equally long
        for (auto &v : node) v.resize(node[0].size());
        for (auto &v : list) v.resize(list[0].size());
        bind();
        A.cur = list[0].data(); A.below = list[1].data();
        A.n_nodes = A.n_edges = A.n_cur = A.n_below = 0;
        A.overflow = 0;
    }
    // more room for one kind of array, in the middle of a construction: contents stay, the arena is re-pointed
    bool grow(int which)
    {
        auto wider = [](std::vector<int32_t> &v) { if (v.size() > (size_t)1 << 29) return false; v.resize(v.size() * 2); return true; };
        if (which == malcs_core::GROW_NODES) { for (auto &v : node) if (!wider(v)) return false; }
        else if (which == malcs_core::GROW_EDGES) { if (edge.size() > (size_t)6 << 29 || !wider(edge)) return false; }
        else {
            const bool cur_is_0 = A.cur == list[0].data();
            for (auto &v : list) if (!wider(v)) return false;
            A.cur = list[cur_is_0 ? 0 : 1].data(); A.below = list[cur_is_0 ? 1 : 0].data();
        }
        bind();
        return true;
    }
};

// builders outlive the call that made them: a worker thread takes one and gives it back
struct builder_pool {
    std::mutex lock;
    std::vector<std::unique_ptr<builder>> spare;
    std::unique_ptr<builder> get()
    {
        std::lock_guard<std::mutex> g(lock);
        if (spare.empty()) return std::make_unique<builder>();
        std::unique_ptr<builder> b = std::move(spare.back());
        spare.pop_back();
        return b;
    }
    void put(std::unique_ptr<builder> b) { std::lock_guard<std::mutex> g(lock); if (b && spare.size() < 64) spare.push_back(std::move(b)); }
};
builder_pool &spare_builders() { static builder_pool p; return p; }

// The copying of a large graph's shared edge prefixes (write_graph in two parts, malcs_graph_core.hpp), open to every
// worker thread of the call: a tandem-repeat window's 900 000 edges are 50 000 pool entries, so building it is a
// millisecond and writing it four, and the thread that got the largest graph of a batch would finish long after the
// others.  Nodes are handed out in chunks through `next`.
struct copy_job {
    int32_t *edges = nullptr;
    std::vector<int32_t> at, pre, from;
    int32_t n_nodes = 0;
    std::atomic<int32_t> next{0};
    static constexpr int32_t CHUNK = 64;
    bool help()                                              // one chunk; false: none left to take
    {
        const int32_t lo = next.fetch_add(CHUNK);
        if (lo >= n_nodes) return false;
        malcs_core::copy_prefixes<int32_t>(edges, at.data(), pre.data(), from.data(), lo, std::min(n_nodes, lo + CHUNK));
        return true;
    }
};
struct copy_jobs {
    std::mutex lock;
    std::vector<std::shared_ptr<copy_job>> open;
    std::atomic<int> builders{0};                            // threads that may still add a job
    void add(const std::shared_ptr<copy_job> &j) { std::lock_guard<std::mutex> g(lock); open.push_back(j); }
    std::shared_ptr<copy_job> any()
    {
        std::lock_guard<std::mutex> g(lock);
        while (!open.empty() && open.back()->next.load() >= open.back()->n_nodes) open.pop_back();
        return open.empty() ? nullptr : open.back();
    }
    // after a thread's own pairs: copy for the others until nobody can add a job and none is open
    void drain()
    {
        builders.fetch_sub(1);
        for (;;) {
            if (std::shared_ptr<copy_job> j = any()) { while (j->help()) {} continue; }
            if (builders.load() <= 0) return;
            std::this_thread::yield();
        }
    }
};
// graphs from this size on share their copying (MALCS_SHARED_COPY_EDGES moves the border: tests run both ways)
int64_t shared_copy_edges()
{
    static const int64_t v = [] { const char *e = getenv("MALCS_SHARED_COPY_EDGES"); return e && *e ? atoll(e) : 100000ll; }();
    return v;
}

void build_graph(builder &B, i64_store &store, i32_store &narrow, int64_t m, int64_t n, int64_t shift, const std::vector<run> &runs,
                 int64_t n_levels, pair_graph &G, copy_jobs *jobs = nullptr)
{
    const bool degenerate = runs.empty() || n_levels == 0;
    if (degenerate) {                                                    // :370-377
        if (m == 0 && n == 0) {
            G.supremal[0] = G.supremal[1] = G.supremal[2] = G.supremal[3] = 0;
            G.nodes = store.take(3);
            G.nodes[0] = shift; G.nodes[1] = shift; G.nodes[2] = 0;
            G.edges = store.take(0);
        } else {
            G.nodes = store.take(6);                                     // the sink keeps unshifted coordinates
            G.nodes[0] = shift; G.nodes[1] = shift; G.nodes[2] = 0; G.nodes[3] = m; G.nodes[4] = n; G.nodes[5] = 0;
            G.edges = store.take(6);
            G.edges[0] = 0; G.edges[1] = 1; G.edges[2] = shift; G.edges[3] = shift + m; G.edges[4] = 0; G.edges[5] = n;
            G.supremal[0] = shift; G.supremal[1] = shift + m; G.supremal[2] = 0; G.supremal[3] = n;
        }
        return;
    }
    const int32_t R = (int32_t)runs.size();
    B.runs32.resize((size_t)R * 4);
    for (int32_t r = 0; r < R; r++) {
        B.runs32[4 * r] = (int32_t)runs[r].row; B.runs32[4 * r + 1] = (int32_t)runs[r].col;
        B.runs32[4 * r + 2] = (int32_t)runs[r].len; B.runs32[4 * r + 3] = (int32_t)runs[r].level;
    }
    B.start(2 * R + 16, 4 * R + 64, R + 8);
    malcs_core::arena &A = B.A;
    int32_t source = 0, sink = 0, sup[4] = {0, 0, 0, 0};
    malcs_core::build(A, (int32_t)m, (int32_t)n, shift, B.runs32.data(), R, (int32_t)n_levels, source, sink, sup,
                      [&B](malcs_core::arena &, int which) { return B.grow(which); });
    if (A.overflow) throw std::bad_alloc();                              // an array beyond 2^30 entries
    int64_t n_edges = 0;
    const int32_t count = malcs_core::count_reachable(A, source, n_edges);
    if (n_edges > 0x7fffffffll / 6) throw std::bad_alloc();
    int32_t *words = narrow.take((size_t)count * 3 + (size_t)n_edges * 6);
    if (B.where.size() < (size_t)A.n_edges) B.where.resize((size_t)A.n_edges);
    if (jobs && n_edges >= shared_copy_edges()) {
        std::shared_ptr<copy_job> j = std::make_shared<copy_job>();
        j->edges = words + (size_t)count * 3;
        j->n_nodes = count;
        j->at.resize((size_t)count); j->pre.resize((size_t)count); j->from.resize((size_t)count);
        malcs_core::write_graph_own<int32_t>(A, source, words, j->edges, 0, B.where.data(), j->at.data(), j->pre.data(), j->from.data());
        jobs->add(j);
        while (j->help()) {}                                 // with whoever else is free; chunks taken are finished before the call returns
    } else {
        malcs_core::write_graph<int32_t>(A, source, words, words + (size_t)count * 3, 0, B.where.data());
    }
    G.narrow = true;
    G.dev_nodes = words; G.dev_edges = words + (size_t)count * 3;
    G.dev_n_nodes = count; G.dev_n_edges = n_edges; G.dev_shift = shift;
    G.supremal[0] = sup[0] + shift; G.supremal[1] = sup[1] + shift; G.supremal[2] = sup[2]; G.supremal[3] = sup[3];
    // a thread that has built one graph of a million edges does not keep 100 MB for the small ones that follow
    if (B.edge.size() > ((size_t)6 << 22)) { B.edge.resize((size_t)6 << 22); B.edge.shrink_to_fit(); }
    if (B.where.size() > ((size_t)1 << 22)) { B.where.resize((size_t)1 << 22); B.where.shrink_to_fit(); }
    if (B.node[0].size() > ((size_t)1 << 20)) { for (auto &v : B.node) { v.resize((size_t)1 << 20); v.shrink_to_fit(); } for (auto &v : B.list) { v.resize((size_t)1 << 20); v.shrink_to_fit(); } }
}

// ---- canonical allele ---------------------------------------------------------------------------
// Restatement of the reference's canonical() (algebra/extractor/extractor.py:10-89) on the node / edge
// arrays: BFS from the source; a node reached again at the same depth merges the two ways in into one
// delins up to their lowest common ancestor; finally walk back from the sink.  Nodes are keyed by value
// (row, col, length), as the reference's dict does.
struct cvar { int64_t start, end, from, to; };      // variant; sequence = supremal.sequence[from:to] or window slice (see below)

void canonical_of(pair_graph &G)
{
    const size_t nn = G.nodes.size() / 3, ne = G.edges.size() / 6;
    G.canon.clear();
    G.canon_done = true;
    if (nn == 0) return;
    // CSR over edges (already grouped by source in node order)
    std::vector<size_t> first(nn + 1, 0);
    for (size_t e = 0; e < ne; e++) first[(size_t)G.edges[6 * e] + 1]++;
    for (size_t v = 0; v < nn; v++) first[v + 1] += first[v];
    // value -> representative id
    std::vector<int32_t> rep(nn);
    {
        std::vector<int32_t> order(nn);
        for (size_t v = 0; v < nn; v++) order[v] = (int32_t)v;
        auto key_less = [&](int32_t a, int32_t b) {
            for (int k = 0; k < 3; k++) if (G.nodes[3 * a + k] != G.nodes[3 * b + k]) return G.nodes[3 * a + k] < G.nodes[3 * b + k];
            return a < b;
        };
        std::sort(order.begin(), order.end(), key_less);
        for (size_t i = 0; i < nn; i++) {
            const int32_t v = order[i];
            const bool same = i && G.nodes[3 * order[i - 1]] == G.nodes[3 * v] && G.nodes[3 * order[i - 1] + 1] == G.nodes[3 * v + 1] &&
                              G.nodes[3 * order[i - 1] + 2] == G.nodes[3 * v + 2];
            rep[v] = same ? rep[order[i - 1]] : v;
        }
    }
    const int64_t shift = G.nodes[0];                        // source.row
    const int64_t sup_from = G.supremal[2], sup_to = G.supremal[3];
    auto row = [&](int32_t v) { return G.nodes[3 * v]; };
    auto col = [&](int32_t v) { return G.nodes[3 * v + 1]; };
    auto len = [&](int32_t v) { return G.nodes[3 * v + 2]; };
    // slice of supremal.sequence -> slice of the observed window (Python slice clamping)
    auto spelled = [&](int32_t anchor, int64_t start, int32_t target, int64_t end) -> cvar {
        int64_t x = col(anchor) + start - row(anchor) - shift, y = col(target) + end - row(target) - shift;
        const int64_t n = sup_to - sup_from;
        x = std::max<int64_t>(0, std::min(x, n)); y = std::max<int64_t>(0, std::min(y, n));
        if (y < x) y = x;
        return {start, end, sup_from + x, sup_from + y};
    };
    struct way { int32_t parent; cvar edge; int64_t depth; bool has_edge; };
    std::vector<way> best(nn);
    std::vector<uint8_t> seen(nn, 0);
    struct item { int32_t node, parent; cvar edge; int64_t depth; bool has_edge; };
    std::vector<item> queue;
    queue.push_back({0, -1, {0, 0, 0, 0}, 0, false});
    int32_t sink = -1;
    for (size_t head = 0; head < queue.size(); head++) {
        const item it = queue[head];
        const int32_t node = rep[it.node];
        if (first[it.node + 1] == first[it.node]) sink = node;
        if (!seen[node]) {
            seen[node] = 1;
            best[node] = {it.parent < 0 ? -1 : rep[it.parent], it.edge, it.depth, it.has_edge};
            for (size_t e = first[it.node]; e < first[it.node + 1]; e++) {
                const int64_t *x = &G.edges[6 * e];
                queue.push_back({(int32_t)x[1], it.node, {x[2], x[3], x[4], x[5]}, it.depth + 1, true});
            }
            continue;
        }
        const way old = best[node];
        if (it.depth > old.depth) continue;
        // lowest common ancestor of the two ways in; the right-hand edge is not reset between outer rounds (:33-40)
        int32_t left = old.parent, right = rep[it.parent];
        cvar left_edge = old.edge, right_edge = it.edge;
        int32_t ancestor = -1;
        while (left >= 0 && ancestor < 0) {
            int32_t probe = right;
            while (probe >= 0) {
                if (probe == left) { ancestor = left; break; }
                right_edge = best[probe].edge;
                probe = best[probe].parent;
            }
            if (ancestor >= 0) break;
            left_edge = best[left].edge;
            left = best[left].parent;
        }
        if (ancestor < 0) continue;                           // cannot happen on graphs built by build_graph
        const int64_t start = std::min(left_edge.start, right_edge.start);
        const int32_t parent = rep[it.parent];
        if (old.parent != parent && row(old.parent) + len(old.parent) == row(parent) + len(parent) &&
            col(old.parent) + len(old.parent) == col(parent) + len(parent)) {                         // :63-69
            const int64_t end = std::max(best[parent].edge.end, best[old.parent].edge.end);
            best[old.parent] = {ancestor, spelled(ancestor, start, parent, end), it.depth - 1, true};
        } else {                                                                                   // :70-74
            const int64_t end = std::max(old.edge.end, it.edge.end);
            best[node] = {ancestor, spelled(ancestor, start, node, end), it.depth, true};
        }
    }
    std::vector<cvar> back;
    for (int32_t v = sink; v >= 0;) {                         // :82-89
        const way &w = best[v];
        if (w.parent < 0) break;
        back.push_back(w.edge);
        v = w.parent;
    }
    for (size_t i = back.size(); i-- > 0;) {
        G.canon.push_back(back[i].start); G.canon.push_back(back[i].end);
        G.canon.push_back(back[i].from); G.canon.push_back(back[i].to);
    }
}

struct launch_class { int W, logL; bool band; std::vector<uint32_t> pairs; };

template <typename T> struct dev_buf {               // device memory of a call, from the context's pool
    malcs_ctx *c;
    T *p = nullptr;
    explicit dev_buf(malcs_ctx *ctx) : c(ctx) {}
    dev_buf(const dev_buf &) = delete;
    dev_buf &operator=(const dev_buf &) = delete;
    ~dev_buf() { release(); }
    void release() { if (p) { malcs_pool_free(c, p); p = nullptr; } }
    int alloc(size_t n) { release(); return malcs_pool_alloc(c, (void **)&p, (n ? n : 1) * sizeof(T)) == hipSuccess ? 0 : -1; }
};

uint32_t pow2ceil(uint32_t x) { uint32_t p = 1; while (p < x) p <<= 1; return p; }

double now_ms()
{
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

}  // namespace

// [0]: long windows (workgroup of 512), [1]: windows below 4 096 rows at small distance (a wave each),
// [2], [3]: sequences beyond a resident pattern, whose sweeps also find the distance
struct sweep_set { std::vector<uint32_t> pairs; pinned_vec<uint32_t> flat; std::vector<uint64_t> first; };

struct malcs_graphs {
    std::vector<pair_graph> g;
    sweep_set sweeps[4];
    std::deque<i64_store> stores;   // one per worker thread
    std::deque<i32_store> narrow;   // likewise: the int32 arrays of the graphs built on the host
    std::deque<pinned_vec<int32_t>> dev_words;    // graphs the device built, one buffer per chunk of pairs
    std::deque<pinned_vec<uint32_t>> bv_runs;     // ... and, for the bit-vector paths, their runs (the sweeps keep theirs in `flat`)
    float timing[4] = {0, 0, 0, 0};               // [3]: device graph construction (malcs_gbuild.hip)
    uint64_t n_dev_built = 0;
    uint64_t tie_bytes = 0, tie_pairs = 0;        // bit-vector path: bytes of tie-bit rows its forward pass wrote (and its backward pass read), pairs
};

// the int64 arrays of a graph the device built (rows, columns and variant positions get their shift here)
static void widen(malcs_graphs *g, pair_graph &G)
{
    if (!G.narrow || G.nodes.p) return;
    if (g->stores.empty()) g->stores.resize(1);
    if (G.dev_built) {
        const uint32_t *src = G.dev_runs_in ? g->sweeps[G.dev_runs_in - 1].flat.data() + 4 * G.dev_runs_at : G.dev_runs;
        G.runs = g->stores[0].take((size_t)G.dev_n_runs * 4);
        for (int64_t r = 0; r < G.dev_n_runs; r++) {
            G.runs[4 * r] = src[4 * r] + G.dev_shift; G.runs[4 * r + 1] = src[4 * r + 1] + G.dev_shift;
            G.runs[4 * r + 2] = src[4 * r + 2]; G.runs[4 * r + 3] = src[4 * r + 3];
        }
    }
    G.nodes = g->stores[0].take((size_t)G.dev_n_nodes * 3);
    G.edges = g->stores[0].take((size_t)G.dev_n_edges * 6);
    for (int64_t v = 0; v < G.dev_n_nodes; v++) {
        G.nodes[3 * v] = G.node_word(3 * v) + G.dev_shift; G.nodes[3 * v + 1] = G.node_word(3 * v + 1) + G.dev_shift;
        G.nodes[3 * v + 2] = G.node_word(3 * v + 2);
    }
    for (int64_t e = 0; e < G.dev_n_edges; e++) {
        int64_t *o = &G.edges[6 * e];
        o[0] = G.edge_word(6 * e); o[1] = G.edge_word(6 * e + 1); o[2] = G.edge_word(6 * e + 2) + G.dev_shift;
        o[3] = G.edge_word(6 * e + 3) + G.dev_shift; o[4] = G.edge_word(6 * e + 4); o[5] = G.edge_word(6 * e + 5);
    }
}

// After the sweeps have compacted a chunk's runs on the device: order them and build the graphs there
// (malcs_gbuild.hip), bring the finished node / edge arrays home.  Pairs the kernels give up stay as they are (their
// runs, now sorted, come home with the others) and are built by the host half.
struct gbuild_user {
    malcs_batch *b; malcs_graphs *out; const int64_t *shift; const int64_t *d_shift;
    const std::vector<uint32_t> *work;                                   // the work list the chunk's work_index refers to
    int set;                                                             // the sweep set whose flat array takes the chunk's runs (-1: the caller points the handles at them)
};

static int gbuild_hook(void *user_, const malcs_sweep_chunk &c)
{
    gbuild_user &u = *static_cast<gbuild_user *>(user_);
    malcs_stage_trace stages("graph kernels");
    malcs_ctx *ctx = u.b->ctx;
    hipStream_t st = ctx->stream;
    const int PO = malcs_gbuild_pair_out_words();
    std::vector<uint32_t> tiny, small, big;
    std::vector<uint64_t> scratch_off(c.count, 0);
    uint64_t scratch_words = 0, out_cap = 0;
    for (uint32_t f = 0; f < c.count; f++) {
        const uint32_t R = c.cnt[f], pid = c.pids[f];
        if (R == 0 || u.b->a_len[pid] >= (1u << 30) || u.b->b_len[pid] >= (1u << 30)) continue;
        if (R <= 32) tiny.push_back(f);
        else if (R <= 64) small.push_back(f);
        else if (R <= 1024) { big.push_back(f); scratch_off[f] = scratch_words; scratch_words += (uint64_t)malcs_core::arena_words(2 * (int32_t)R + 16, 8 * (int32_t)R + 64, (int32_t)R + 8); }
        else continue;
        out_cap += 3ull * (2ull * R + 8) + 6ull * (4ull * R + 16);      // more than graphs of R runs usually need; beyond it the host builds
    }
    if (tiny.empty() && small.empty() && big.empty()) return MALCS_OK;
    const size_t n_slots = tiny.size() + small.size() + big.size();
    pool_scoped<uint32_t> d_slots(ctx);
    pool_scoped<uint64_t> d_scratch_off(ctx);
    pool_scoped<int32_t> d_scratch(ctx), d_words(ctx), d_pair_out(ctx);
    pool_scoped<unsigned long long> d_cursor(ctx);
    MALCS_HIP(d_slots.alloc(n_slots)); MALCS_HIP(d_scratch_off.alloc(c.count)); MALCS_HIP(d_scratch.alloc(std::max<uint64_t>(scratch_words, 1)));
    MALCS_HIP(d_words.alloc(out_cap)); MALCS_HIP(d_pair_out.alloc((size_t)c.count * PO)); MALCS_HIP(d_cursor.alloc(1));
    stages.lap("classes, allocations");
    std::vector<uint32_t> slots(tiny);
    slots.insert(slots.end(), small.begin(), small.end());
    slots.insert(slots.end(), big.begin(), big.end());
    MALCS_HIP(malcs_h2d(st, d_slots.p, slots.data(), n_slots * sizeof(uint32_t)));
    MALCS_HIP(malcs_h2d(st, d_scratch_off.p, scratch_off.data(), (size_t)c.count * sizeof(uint64_t)));
    MALCS_HIP(hipMemsetAsync(d_cursor.p, 0, sizeof(unsigned long long), st));
    // The kernels run group by group (the graphs of at most 32 runs in up to four groups, then those of up to 64, then the large ones) and say after every
    // group how far the output has grown: while a group is being built, the words of the one before travel home on the
    // context's second stream (104 MB for 100 000 C2 graphs: 1.9 ms that used to follow the 2.5 ms of the kernels).
    if (!ctx->copy_stream) MALCS_HIP(hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
    // Whatever way this function is left, both streams are drained first: the pooled device blocks above go back to the
    // context's pool when it returns, and the pool only promises that a block is handed out again behind work queued on
    // the context's stream -- copies on the second stream (and their pinned targets) must be done by then.
    struct drain {
        hipStream_t a, b;
        ~drain() { (void)hipStreamSynchronize(a); (void)hipStreamSynchronize(b); }
    } drain_on_exit{st, ctx->copy_stream};
    struct group { size_t first, count; int kind; };
    std::vector<group> groups;
    {
        const size_t parts = tiny.size() >= 4096 ? 4 : 1, per = (tiny.size() + parts - 1) / parts;
        for (size_t at = 0; at < tiny.size(); at += per) groups.push_back({at, std::min(per, tiny.size() - at), 0});
        if (!small.empty()) groups.push_back({tiny.size(), small.size(), 1});
        if (!big.empty()) groups.push_back({tiny.size() + small.size(), big.size(), 2});
    }
    struct event_set {
        std::vector<hipEvent_t> ev;
        ~event_set() { for (hipEvent_t e : ev) (void)hipEventDestroy(e); }
    } done, ordered;
    pinned_vec<unsigned long long> grown;
    grown.resize(groups.size());
    MALCS_HIP(hipEventRecord(ctx->ev1, st));
    // the runs in the reference's order first, every slot's in place: they are final then (what the handle exposes, and
    // what the host half builds from if a kernel gives a pair up) and go home on the second stream, under the construction
    {
        int rc = malcs_gbuild_order(ctx, 0, c.d_runs, c.d_run_off, c.d_run_cnt, c.d_pids, d_slots.p, (uint32_t)(tiny.size() + small.size()),
                                    u.b->d_a_len, u.b->d_b_len);
        if (!rc) rc = malcs_gbuild_order(ctx, 2, c.d_runs, c.d_run_off, c.d_run_cnt, c.d_pids, d_slots.p + tiny.size() + small.size(),
                                         (uint32_t)big.size(), u.b->d_a_len, u.b->d_b_len);
        if (rc) return rc;
        if (c.host_runs && c.n_runs && c.runs_home) {
            hipEvent_t e = nullptr;
            MALCS_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            ordered.ev.push_back(e);
            MALCS_HIP(hipEventRecord(e, st));
            MALCS_HIP(hipStreamWaitEvent(ctx->copy_stream, e, 0));
            MALCS_HIP(hipMemcpyAsync(c.host_runs, c.d_runs, c.n_runs * 16ull, hipMemcpyDeviceToHost, ctx->copy_stream));
            *c.runs_home = true;                                          // (both streams are drained before this function returns)
        }
    }
    for (size_t k = 0; k < groups.size(); k++) {
        const int rc = malcs_gbuild_launch(ctx, groups[k].kind, c.d_runs, c.d_run_off, c.d_run_cnt, c.d_pids, d_slots.p + groups[k].first,
                                           (uint32_t)groups[k].count, u.b->d_a_len, u.b->d_b_len, u.b->d_dist, u.d_shift, d_scratch.p,
                                           d_scratch_off.p, d_words.p, out_cap, d_cursor.p, d_pair_out.p);
        if (rc) return rc;
        MALCS_HIP(hipMemcpyAsync(&grown.data()[k], d_cursor.p, sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
        hipEvent_t e = nullptr;
        MALCS_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        done.ev.push_back(e);
        MALCS_HIP(hipEventRecord(e, st));
    }
    MALCS_HIP(hipEventRecord(ctx->ev2, st));
    pinned_vec<int32_t> pair_out;
    pair_out.resize((size_t)c.count * PO);
    MALCS_HIP(hipMemcpyAsync(pair_out.data(), d_pair_out.p, pair_out.size() * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    // group k's words are [reach[k], reach[k + 1]) of the device buffer and all of home[k]
    std::vector<uint64_t> reach(groups.size() + 1, 0);
    std::vector<const int32_t *> home(groups.size(), nullptr);
    for (size_t k = 0; k < groups.size(); k++) {
        MALCS_HIP(hipEventSynchronize(done.ev[k]));
        reach[k + 1] = std::max<uint64_t>(reach[k], std::min<unsigned long long>(grown.data()[k], out_cap));
        const uint64_t words_k = reach[k + 1] - reach[k];
        u.out->dev_words.emplace_back();
        pinned_vec<int32_t> &words = u.out->dev_words.back();
        words.resize((size_t)words_k);
        home[k] = words.data();
        if (words_k)
            MALCS_HIP(hipMemcpyAsync(words.data(), d_words.p + reach[k], (size_t)words_k * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->copy_stream));
    }
    MALCS_HIP(malcs_sync(st));
    float ms = 0;
    MALCS_HIP(hipEventElapsedTime(&ms, ctx->ev1, ctx->ev2));
    u.out->timing[3] += ms;
    stages.lap("kernels, counts home");
    MALCS_HIP(hipStreamSynchronize(ctx->copy_stream));
    stages.lap("graphs home");
    // the handles of the pairs that were built point at what came home: a loop over the pairs whose cost is the cache lines
    // of handle it touches per pair (0.4 ms for 100 000 on one thread), so the context's threads share large chunks
    std::atomic<uint64_t> built(0);
    auto point_at = [&](size_t k, size_t q_lo, size_t q_hi) {
        uint64_t mine = 0;
        for (size_t q = q_lo; q < q_hi; q++) {
            const uint32_t f = slots[q];
            const int32_t *po = pair_out.data() + (size_t)f * PO;
            if (po[0] != 0) continue;                                     // given up: the host half builds this one
            const uint64_t at = (uint64_t)(uint32_t)po[7] | ((uint64_t)(uint32_t)po[8] << 32);
            const uint64_t n_nodes = (uint64_t)po[1], n_edges = (uint64_t)po[2];
            const bool half16 = (po[9] >> 30) & 1;                            // node / edge words as int16
            const uint64_t items = 3 * n_nodes + 6 * n_edges;
            if (at < reach[k] || at + (half16 ? (items + 1) / 2 : items) > reach[k + 1]) continue;
            pair_graph &G = u.out->g[c.pids[f]];
            const int64_t sh = u.shift ? u.shift[c.pids[f]] : 0;
            const int32_t *words = home[k] + (at - reach[k]);
            if (half16) { G.dev_nodes16 = reinterpret_cast<const int16_t *>(words); G.dev_edges16 = G.dev_nodes16 + 3 * n_nodes; }
            else { G.dev_nodes = words; G.dev_edges = words + 3 * n_nodes; }
            G.dev_n_nodes = (int64_t)n_nodes; G.dev_n_edges = (int64_t)n_edges; G.dev_shift = sh;
            G.dev_n_runs = (int64_t)c.cnt[f];
            G.n_levels = ((int64_t)u.b->a_len[c.pids[f]] + (int64_t)u.b->b_len[c.pids[f]] - G.distance) / 2;
            if (u.set >= 0) { G.dev_runs_in = (uint8_t)(u.set + 1); G.dev_runs_at = c.runs_base + c.run_off[f]; }
            G.supremal[0] = po[3] + sh; G.supremal[1] = po[4] + sh; G.supremal[2] = po[5]; G.supremal[3] = po[6];
            G.dev_built = true;
            G.narrow = true;
            mine++;
        }
        built += mine;
    };
    malcs_workers *wk = n_slots >= 16384 ? malcs_ctx_workers(ctx) : nullptr;
    const int nt = wk ? (int)std::min<size_t>(16, wk->size()) : 1;
    for (size_t k = 0; k < groups.size(); k++) {
        const size_t lo = groups[k].first, cnt = groups[k].count;
        if (nt > 1 && cnt >= 4096) wk->run(nt, [&](int t) { point_at(k, lo + cnt * (size_t)t / (size_t)nt, lo + cnt * (size_t)(t + 1) / (size_t)nt); });
        else point_at(k, lo, lo + cnt);
    }
    u.out->n_dev_built += built.load();
    stages.lap("handles");
    return MALCS_OK;
}

static int graph_batch(malcs_batch *b, const int64_t *shift, const int64_t *max_distance, malcs_graphs *out)
{
    malcs_ctx *ctx = b->ctx;
    hipStream_t st = ctx->stream;
    const uint64_t N = b->n_pairs;
    malcs_stage_trace stages("lcs_graph_batch");
    out->g.resize(N);
    std::vector<event_span> pair_events(N);
    std::deque<std::vector<event>> event_store;          // one buffer per chunk, alive until the graphs are built
    std::deque<pinned_vec<uint32_t>> &bv_runs = out->bv_runs;   // the sorted runs of the pairs of a chunk the device built from its events (kept by the handle)
    std::vector<uint8_t> has_runs(N, 0);

    // [0]: long windows (workgroup of 512), [1]: windows below 4 096 rows at small distance (a wave each),
    // [2], [3]: sequences beyond a resident pattern, whose sweeps also find the distance
    sweep_set (&sweeps)[4] = out->sweeps;                // the handle keeps them: runs are read in place
    std::vector<uint64_t> sweep_slot(N, 0);
    int64_t sweep_dmax = 32768;                         // beyond: the table (D^2 / 2 words) and the single workgroup stop paying
    if (const char *e = getenv("MALCS_SWEEP_DMAX")) sweep_dmax = atoll(e);
    float ms_fwd = 0, ms_reach = 0;

    // exact distances first (kernel family 1): they decide max_distance, the band and the event budget
    std::vector<int32_t> dist(N, -1);
    stages.lap("handles and lists made");
    if (N) {
        int rc = malcs_dist_launch_resident(b);
        if (rc) return rc;
        stages.lap("distance kernels queued");
        MALCS_HIP(malcs_d2h(st, dist.data(), b->d_dist, N * 4ull));
        MALCS_HIP(malcs_sync(st));
        stages.lap("distances home");
        // Sequences beyond a resident pattern: the backward sweep finds the distance while it fills its table, first
        // with room for 1 024 edits, then for sweep_dmax; what is further apart goes to the band kernels.
        std::vector<uint32_t> todo;
        for (uint32_t pid : b->band_pairs) {
            dist[pid] = -1;
            const int64_t md = max_distance ? max_distance[pid] : 0;
            if (b->a_len[pid] >= 4096u && !md) todo.push_back(pid); // with a max_distance the verdict may come before any graph
        }
        const int64_t caps[2] = {std::min<int64_t>(1024, sweep_dmax), sweep_dmax};
        for (int stage = 0; stage < 2 && !todo.empty() && caps[stage] > 0; stage++) {
            if (stage == 1 && caps[1] <= caps[0]) break;
            sweep_set &S = sweeps[2 + stage];
            S.pairs = todo;
            float ms[2] = {0, 0};
            rc = malcs_sweep_optimal_runs(b, false, S.pairs, dist.data(), (int32_t)caps[stage], S.flat, S.first, ms);
            if (rc) return rc;
            ms_fwd += ms[0]; ms_reach += ms[1];
            todo.clear();
            for (size_t s = 0; s < S.pairs.size(); s++) {
                const uint32_t pid = S.pairs[s];
                if (dist[pid] < 0) { todo.push_back(pid); continue; }
                has_runs[pid] = (uint8_t)(4 + stage);
                sweep_slot[pid] = s;
                MALCS_HIP(malcs_h2d(st, b->d_dist + pid, &dist[pid], 4));
            }
        }
        std::vector<uint32_t> rest;
        for (uint32_t pid : b->band_pairs) if (dist[pid] < 0) rest.push_back(pid);
        if (!rest.empty()) {
            std::vector<int32_t> got(rest.size());
            rc = malcs_dist_band(b, rest, got.data());
            if (rc) return rc;
            for (size_t i = 0; i < rest.size(); i++) dist[rest[i]] = got[i];
        }
    }

    {   // the distances into the handles: by the context's threads when the batch is large
        auto put = [&](uint64_t lo, uint64_t hi) { for (uint64_t i = lo; i < hi; i++) out->g[i].distance = dist[i]; };
        malcs_workers *wk = N >= 16384 ? malcs_ctx_workers(ctx) : nullptr;
        const int nt = wk ? (int)std::min<size_t>(16, wk->size()) : 1;
        if (nt > 1) wk->run(nt, [&](int t) { put(N * (uint64_t)t / (uint64_t)nt, N * (uint64_t)(t + 1) / (uint64_t)nt); });
        else put(0, N);
    }
    stages.lap("distances");
    // launch classes.  Short windows keep all their row groups resident (L lanes per pair, several pairs per
    // wave); longer ones get a wave each whose 64 lanes slide along the band given by the distance.
    const int band_wmax = malcs_graph_max_band_words(b->n_planes);
    auto band_capacity = [](int W) { const int G = 64 * W; return 61 * (G + 1) - G - 64; };
    std::vector<launch_class> classes;
    const bool resident_band = !(getenv("MALCS_GRAPH_RESIDENT_BAND") && atoi(getenv("MALCS_GRAPH_RESIDENT_BAND")) == 0);   // 0: resident windows store every group's tie bits
    int64_t small_rows = 192;                           // shorter windows stay with the bit-vector kernels (64 pairs per wave)
    if (const char *e = getenv("MALCS_SWEEP_MIN_ROWS")) small_rows = atoll(e);
    int64_t small_ratio = 8;                            // ... as do windows further apart than length / 8
    if (const char *e = getenv("MALCS_SWEEP_RATIO")) small_ratio = atoll(e);
    for (uint64_t i = 0; i < N; i++) {
        const uint32_t m = b->a_len[i], n = b->b_len[i];
        const int64_t D = dist[i];                    // (the loop reads lengths and distances only: a handle is 200 bytes away from the next)
        const int64_t md = max_distance ? max_distance[i] : 0;
        if (md && D > md) { out->g[i].status = MALCS_PAIR_MAX_DISTANCE_EXCEEDED; continue; }
        if (m == 0 || n == 0) continue;               // no DP needed: LLCS = 0
        if (has_runs[i] >= 4) continue;               // its runs came with its distance
        int W = 0, logL = 0;
        bool band = false;
        if (m >= 4096u && D <= sweep_dmax) {  // long and similar: the output-sensitive sweeps (malcs_sweep.hip)
            sweeps[0].pairs.push_back((uint32_t)i);
            has_runs[i] = 2;
            continue;
        }
        if (m < 4096u && (int64_t)m >= small_rows && D <= std::min<int64_t>(sweep_dmax, malcs_sweep_small_dmax()) &&
            small_ratio * D <= (int64_t)std::max(m, n)) {   // similar windows: D + 1 steps of a wave instead of n columns
            sweeps[1].pairs.push_back((uint32_t)i);
            has_runs[i] = 3;
            continue;
        }
        if (m < 4096u) {
            const uint32_t words = m / 64u + 1u;
            if (resident_band && words >= 4 && 4 * (D + 2) <= (int64_t)m) {
                // all groups resident, but only the band's tie bits stored (the distance is known: the band kernels' bounds):
                // groups of 64 rows, so that the two or three groups a column's band touches are a fraction of the window
                W = 1;
                const uint32_t lanes = pow2ceil(words);
                while ((1u << logL) < lanes) logL++;
            } else if (words <= 4) W = (int)pow2ceil(words);
            else { W = 4; const uint32_t lanes = pow2ceil((words + 3u) / 4u); while ((1u << logL) < lanes) logL++; }
        } else {
            band = true;
            logL = 6;
            for (W = 1; W <= band_wmax && band_capacity(W) < D + 2; W *= 2) {}
            if (W > band_wmax) {
                if (m < 16384u) {                       // too divergent for a band: all groups resident
                    band = false; W = 4; logL = 0;
                    const uint32_t lanes = pow2ceil((m / 64u + 1u + 3u) / 4u);
                    while ((1u << logL) < lanes) logL++;
                } else {
                    malcs_set_error("pair %llu: windows of %u and %u symbols at distance %lld exceed the widest band "
                                    "(%d edits) the graph kernels hold", (unsigned long long)i, m, n, (long long)D,
                                    band_capacity(band_wmax) - 2);
                    return MALCS_ERR_TOO_LONG;
                }
            }
        }
        size_t c = 0;
        for (; c < classes.size(); c++) if (classes[c].W == W && classes[c].logL == logL && classes[c].band == band) break;
        if (c == classes.size()) classes.push_back({W, logL, band, {}});
        classes[c].pairs.push_back((uint32_t)i);
        has_runs[i] = 1;
    }

    size_t free_b = 0, total_b = 0;
    if (!classes.empty()) MALCS_HIP(hipMemGetInfo(&free_b, &total_b));     // (only the bit-vector classes size their chunks by it; the call is 0.2 ms)
    const uint64_t budget_dwords = std::min<uint64_t>(free_b / 2, 64ull << 30) / 4;

    stages.lap("launch classes");
    pool_scoped<int64_t> d_shift(ctx);                                   // the shifts, for the graph kernels (the trim of a sink no edge reaches needs them)
    const bool on_device = !(getenv("MALCS_GBUILD") && atoi(getenv("MALCS_GBUILD")) == 0);      // MALCS_GBUILD=0: host half only
    if (shift && N && on_device && (!sweeps[0].pairs.empty() || !sweeps[1].pairs.empty() || !classes.empty())) {
        MALCS_HIP(d_shift.alloc(N));
        MALCS_HIP(hipMemcpyAsync(d_shift.p, shift, N * sizeof(int64_t), hipMemcpyHostToDevice, st));
        MALCS_HIP(malcs_sync(st));
    }
    {   // the one-wave sweeps take more pairs per wave the closer the pairs: by that number, if the list mixes them
        std::vector<uint32_t> &P = sweeps[1].pairs;
        bool mixed = false;
        for (size_t s = 1; s < P.size() && !mixed; s++) mixed = malcs_sweep_pairs_per_wave(dist[P[s]]) != malcs_sweep_pairs_per_wave(dist[P[0]]);
        if (mixed)
            std::stable_sort(P.begin(), P.end(), [&](uint32_t x, uint32_t y) { return malcs_sweep_pairs_per_wave(dist[x]) > malcs_sweep_pairs_per_wave(dist[y]); });
    }
    for (int shape = 0; shape < 2; shape++) {
        sweep_set &S = sweeps[shape];
        if (S.pairs.empty()) continue;
        float ms[2] = {0, 0};
        gbuild_user user{b, out, shift, d_shift.p, &S.pairs, shape};
        int rc = malcs_sweep_optimal_runs(b, shape == 1, S.pairs, dist.data(), 0, S.flat, S.first, ms, on_device ? gbuild_hook : nullptr, &user);
        if (rc) return rc;
        ms_fwd += ms[0]; ms_reach += ms[1];
        for (size_t s = 0; s < S.pairs.size(); s++) sweep_slot[S.pairs[s]] = s;
    }
    stages.lap("sweeps");
    dev_buf<uint32_t> d_work(b->ctx), d_evcap(b->ctx), d_evcnt(b->ctx), d_scratch(b->ctx);
    dev_buf<uint64_t> d_scoff(b->ctx), d_evoff(b->ctx), d_outoff(b->ctx);
    dev_buf<int32_t> d_kmin(b->ctx), d_kmax(b->ctx), d_rw(b->ctx), d_llcs(b->ctx), d_fdist(b->ctx);
    dev_buf<uint8_t> d_skip(b->ctx);
    dev_buf<event> d_events(b->ctx), d_packed(b->ctx);

    for (const launch_class &lc : classes) {
        const uint64_t NL = 2ull * lc.W;
        const int logG = lc.W == 1 ? 6 : (lc.W == 2 ? 7 : (lc.W == 4 ? 8 : 9));
        size_t pos = 0;
        while (pos < lc.pairs.size()) {
            // ---- chunk of pairs whose tie-bit rows fit the scratch budget ----
            std::vector<uint32_t> work;
            std::vector<uint64_t> sc_off;
            std::vector<int32_t> kmin, kmax, rw, llcs;
            uint64_t used = 0;
            while (pos < lc.pairs.size()) {
                const uint32_t pid = lc.pairs[pos];
                const int64_t m = b->a_len[pid], n = b->b_len[pid], D = dist[pid];
                int32_t lo = -(1 << 30), hi = 1 << 30, groups = 1 << lc.logL;
                if (lc.band || resident_band) {
                    const int64_t delta = n - m;
                    lo = (int32_t)(-((D - delta) / 2) - 1);
                    hi = (int32_t)((D + delta) / 2 + 1);
                    const int32_t in_band = (int32_t)(((int64_t)hi - lo) >> logG) + 2;
                    groups = lc.band ? in_band : std::min(groups, in_band);        // (resident: never more than the window has)
                }
                // columns 0 ... n, one record per group in the band; resident windows index by systolic step (column + group)
                const uint64_t need = ((uint64_t)n + 1ull + (lc.band ? 0ull : (uint64_t)groups)) * (uint64_t)groups * 2ull * NL;
                if (!work.empty() && used + need > budget_dwords) break;
                if (need > budget_dwords) {
                    malcs_set_error("pair %u needs %llu MiB of tie-bit rows; device has %llu MiB free", pid,
                                    (unsigned long long)(need >> 18), (unsigned long long)(free_b >> 20));
                    return MALCS_ERR_NOMEM;
                }
                work.push_back(pid); sc_off.push_back(used); used += need; pos++;
                kmin.push_back(lo); kmax.push_back(hi); rw.push_back(groups);
                llcs.push_back((int32_t)((m + n - D) / 2));
            }
            const uint32_t cnt = (uint32_t)work.size();
            stages.lap("  chunk lists");
            out->tie_bytes += used * 4ull; out->tie_pairs += cnt;
            if (d_work.alloc(cnt) || d_scoff.alloc(cnt) || d_scratch.alloc(used) || d_skip.alloc(cnt) ||
                d_evcap.alloc(cnt) || d_evcnt.alloc(cnt) || d_evoff.alloc(cnt) || d_outoff.alloc(cnt) ||
                d_kmin.alloc(cnt) || d_kmax.alloc(cnt) || d_rw.alloc(cnt) || d_llcs.alloc(cnt) || d_fdist.alloc(cnt)) {
                malcs_set_error("device allocation failed (chunk of %u pairs, %llu MiB of tie-bit rows)", cnt, (unsigned long long)(used >> 18));
                return MALCS_ERR_NOMEM;
            }
            MALCS_HIP(malcs_h2d(st, d_work.p, work.data(), cnt * 4ull));
            MALCS_HIP(malcs_h2d(st, d_scoff.p, sc_off.data(), cnt * 8ull));
            MALCS_HIP(malcs_h2d(st, d_kmin.p, kmin.data(), cnt * 4ull));
            MALCS_HIP(malcs_h2d(st, d_kmax.p, kmax.data(), cnt * 4ull));
            MALCS_HIP(malcs_h2d(st, d_rw.p, rw.data(), cnt * 4ull));
            MALCS_HIP(malcs_h2d(st, d_llcs.p, llcs.data(), cnt * 4ull));
            stages.lap("  allocations, tables up");
            MALCS_HIP(hipEventRecord(ctx->ev0, st));
            int rc = malcs_graph_launch_forward(b, lc.W, lc.logL, lc.band, d_work.p, cnt, d_scratch.p, d_scoff.p,
                                                d_kmin.p, d_kmax.p, d_rw.p, d_fdist.p);
            if (rc) return rc;
            MALCS_HIP(hipEventRecord(ctx->ev1, st));
            std::vector<int32_t> fdist(cnt);
            MALCS_HIP(malcs_d2h(st, fdist.data(), d_fdist.p, cnt * 4ull));
            MALCS_HIP(malcs_sync(st));
            float ms = 0;
            MALCS_HIP(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
            ms_fwd += ms;
            stages.lap("  forward");

            std::vector<uint8_t> skip(cnt, 0);
            std::vector<uint32_t> ev_cap(cnt);
            std::vector<uint64_t> ev_off(cnt);
            uint64_t ev_total = 0;
            for (uint32_t s = 0; s < cnt; s++) {
                const uint32_t pid = work[s];
                if (fdist[s] != dist[pid]) {
                    malcs_set_error("pair %u: forward pass found distance %d inside the band, expected %d (internal error)",
                                    pid, fdist[s], dist[pid]);
                    return MALCS_ERR_INTERNAL;
                }
                ev_cap[s] = 64u + 16u * (uint32_t)dist[pid];
                ev_off[s] = ev_total;
                ev_total += ev_cap[s];
            }
            for (int attempt = 0;; attempt++) {
                if (d_events.alloc(ev_total)) { malcs_set_error("device allocation failed (%llu events)", (unsigned long long)ev_total); return MALCS_ERR_NOMEM; }
                MALCS_HIP(malcs_h2d(st, d_skip.p, skip.data(), cnt));
                MALCS_HIP(malcs_h2d(st, d_evcap.p, ev_cap.data(), cnt * 4ull));
                MALCS_HIP(malcs_h2d(st, d_evoff.p, ev_off.data(), cnt * 8ull));
                MALCS_HIP(hipMemsetAsync(d_evcnt.p, 0, cnt * 4ull, st));
                MALCS_HIP(hipEventRecord(ctx->ev1, st));
                rc = malcs_graph_launch_reach(b, lc.W, lc.logL, lc.band, d_work.p, cnt, d_scratch.p, d_scoff.p, d_kmin.p,
                                              d_kmax.p, d_rw.p, d_llcs.p, d_skip.p, d_events.p, d_evoff.p, d_evcap.p, d_evcnt.p);
                if (rc) return rc;
                MALCS_HIP(hipEventRecord(ctx->ev2, st));
                std::vector<uint32_t> ev_cnt(cnt);
                MALCS_HIP(malcs_d2h(st, ev_cnt.data(), d_evcnt.p, cnt * 4ull));
                MALCS_HIP(malcs_sync(st));
                MALCS_HIP(hipEventElapsedTime(&ms, ctx->ev1, ctx->ev2));
                ms_reach += ms;
                stages.lap("  reach (+ event buffers)");
                bool overflow = false;
                for (uint32_t s = 0; s < cnt; s++) if (ev_cnt[s] > ev_cap[s]) overflow = true;
                if (overflow) {
                    if (attempt) { malcs_set_error("event buffer overflow persisted after a resize"); return MALCS_ERR_INTERNAL; }
                    ev_total = 0;                      // the counts are exact: size every region to its count and redo the pass
                    for (uint32_t s = 0; s < cnt; s++) { ev_cap[s] = std::max(ev_cap[s], ev_cnt[s]); ev_off[s] = ev_total; ev_total += ev_cap[s]; }
                    continue;
                }
                // gather the used parts and bring them home
                std::vector<uint64_t> out_off(cnt);
                uint64_t packed = 0;
                for (uint32_t s = 0; s < cnt; s++) { out_off[s] = packed; packed += ev_cnt[s]; }
                if (d_packed.alloc(packed)) { malcs_set_error("device allocation failed (%llu packed events)", (unsigned long long)packed); return MALCS_ERR_NOMEM; }
                MALCS_HIP(malcs_h2d(st, d_outoff.p, out_off.data(), cnt * 8ull));
                rc = malcs_graph_launch_compact(ctx, d_events.p, d_evoff.p, d_evcnt.p, d_outoff.p, cnt, d_packed.p);
                if (rc) return rc;
                stages.lap("  compaction queued");
                if (on_device && packed) {
                    // The events become runs on the device (k_events_to_runs) and the graph kernels take it from there, as
                    // they do behind the sweeps; what they give up (more than 1 024 runs, events that do not pair up, full
                    // buffers) stays with the host half below, which has the events.
                    std::vector<uint64_t> run_off(cnt);
                    std::vector<uint32_t> run_cnt(cnt), e2r[2];
                    uint64_t runs_total = 0;
                    for (uint32_t s = 0; s < cnt; s++) {
                        const uint32_t E = ev_cnt[s];
                        const bool fits = E >= 2 && !(E & 1u) && E <= 2048u;
                        run_off[s] = runs_total; run_cnt[s] = fits ? E / 2u : 0u;
                        runs_total += run_cnt[s];
                        if (fits) e2r[E <= 128u ? 0 : 1].push_back(s);
                    }
                    if (runs_total) {
                        dev_buf<uint4> d_runs(b->ctx);
                        dev_buf<uint64_t> d_runoff(b->ctx);
                        dev_buf<uint32_t> d_runcnt(b->ctx), d_e2r(b->ctx);
                        if (d_runs.alloc(runs_total) || d_runoff.alloc(cnt) || d_runcnt.alloc(cnt) || d_e2r.alloc(cnt)) {
                            malcs_set_error("device allocation failed (%llu runs)", (unsigned long long)runs_total); return MALCS_ERR_NOMEM;
                        }
                        std::vector<uint32_t> slots(e2r[0]);
                        slots.insert(slots.end(), e2r[1].begin(), e2r[1].end());
                        MALCS_HIP(malcs_h2d(st, d_runoff.p, run_off.data(), cnt * 8ull));
                        MALCS_HIP(malcs_h2d(st, d_e2r.p, slots.data(), slots.size() * 4ull));
                        MALCS_HIP(hipMemsetAsync(d_runcnt.p, 0, cnt * 4ull, st));
                        rc = malcs_gbuild_events_to_runs(ctx, 0, d_packed.p, d_outoff.p, d_evcnt.p, d_e2r.p, (uint32_t)e2r[0].size(), d_runs.p, d_runoff.p, d_runcnt.p);
                        if (!rc) rc = malcs_gbuild_events_to_runs(ctx, 1, d_packed.p, d_outoff.p, d_evcnt.p, d_e2r.p + e2r[0].size(), (uint32_t)e2r[1].size(),
                                                                  d_runs.p, d_runoff.p, d_runcnt.p);
                        if (rc) return rc;
                        MALCS_HIP(malcs_sync(st));                               // (`slots` and `run_off` are locals the copies read)
                        gbuild_user user{b, out, shift, d_shift.p, nullptr, -1};
                        // the runs, sorted by the graph kernels, for the handle (runs(i)) of the pairs that were built there
                        bv_runs.emplace_back();
                        pinned_vec<uint32_t> &flat = bv_runs.back();
                        flat.resize((size_t)runs_total * 4);
                        bool runs_home = false;
                        const malcs_sweep_chunk chunk{d_runs.p, d_runoff.p, d_runcnt.p, d_work.p, cnt, work.data(), run_cnt.data(), run_off.data(), nullptr,
                                                      flat.data(), runs_total, &runs_home, 0};
                        rc = gbuild_hook(&user, chunk);
                        if (rc) return rc;
                        if (!runs_home) MALCS_HIP(hipMemcpyAsync(flat.data(), d_runs.p, runs_total * 16ull, hipMemcpyDeviceToHost, st));
                        MALCS_HIP(malcs_sync(st));
                        for (uint32_t s = 0; s < cnt; s++) {
                            pair_graph &G = out->g[work[s]];
                            if (!G.dev_built) continue;
                            G.dev_runs = flat.data() + 4 * run_off[s];
                            G.dev_n_runs = (int64_t)run_cnt[s];
                        }
                    }
                }
                stages.lap("  runs and graphs on the device");
                // the events come home for the pairs the device has not built (67 MB per 100 000 C2 pairs otherwise, for nothing)
                bool wanted = false;
                for (uint32_t s = 0; s < cnt && !wanted; s++) wanted = !out->g[work[s]].dev_built;
                if (wanted) {
                    event_store.emplace_back(packed);
                    std::vector<event> &host_ev = event_store.back();
                    if (packed) MALCS_HIP(malcs_d2h(st, host_ev.data(), d_packed.p, packed * sizeof(event)));
                    MALCS_HIP(malcs_sync(st));
                    for (uint32_t s = 0; s < cnt; s++) { pair_events[work[s]].p = host_ev.data() + out_off[s]; pair_events[work[s]].n = ev_cnt[s]; }
                }
                stages.lap("  events home");
                break;
            }
        }
    }
    out->timing[0] = ms_fwd;
    out->timing[1] = ms_reach;
    stages.lap("bit-vector classes");

    // ---- host: runs, ordering, graph compression (threads over pairs) ----
    const double t0 = now_ms();
    std::atomic<uint64_t> next(0);
    std::atomic<int> failed(0);
    std::atomic<uint64_t> ns_runs(0), ns_build(0);
    const bool trace = getenv("MALCS_TRACE") != nullptr;
    copy_jobs jobs;
    std::atomic<int> no_memory(0);
    auto worker = [&](i64_store *mine, i32_store *mine32) {
        // whatever happens to this thread's pairs, it leaves through drain(): the others wait for that
        struct leave { copy_jobs &j; ~leave() { j.drain(); } } on_exit{jobs};
        try {
        i64_store &store = *mine;
        std::vector<run> runs;
        std::unique_ptr<builder> kept = spare_builders().get();       // arenas grown by earlier batches, pages touched
        builder &B = *kept;
        struct give_back { std::unique_ptr<builder> &b; ~give_back() { spare_builders().put(std::move(b)); } } back{kept};
        // pairs are taken in blocks: a block of small graphs costs about what the shared counter does, and a graph with
        // hundreds of thousands of edges ends the block that holds it
        const uint64_t block = N >= 4096 ? 64 : 1;
        uint64_t i = 0, block_end = 0;
        for (;;) {
            if (++i >= block_end) { i = next.fetch_add(block); block_end = std::min<uint64_t>(N, i + block); }
            if (i >= N) break;
            pair_graph &G = out->g[i];
            if (G.status != MALCS_PAIR_OK) continue;
            const int64_t m = b->a_len[i], n = b->b_len[i];
            if (G.dev_built) continue;                                   // all done on the device, the handle points at what came home (gbuild_hook)
            runs.clear();
            const double ta = trace ? now_ms() : 0;
            if (has_runs[i] >= 2) {                                      // runs straight from the sweep kernels
                const uint64_t s = sweep_slot[i];
                const pinned_vec<uint32_t> &flat = sweeps[has_runs[i] - 2].flat;
                const std::vector<uint64_t> &first = sweeps[has_runs[i] - 2].first;
                for (uint64_t r = first[s]; r < first[s + 1]; r++)
                    runs.push_back({flat[4 * r], flat[4 * r + 1], flat[4 * r + 2], flat[4 * r + 3]});
                if (!G.dev_built) order_runs(runs, m, n);                 // the graph kernels leave their runs in order
            } else if (has_runs[i] && !runs_from_events(pair_events[i], m, n, runs)) { failed.store(1); continue; }
            const double tb = trace ? now_ms() : 0;
            const int64_t llcs = (m + n - G.distance) / 2;
            G.n_levels = runs.empty() ? (std::min(m, n) > 0 ? 1 : 0) : llcs;
            const int64_t sh = shift ? shift[i] : 0;
            G.runs = store.take(runs.size() * 4);
            for (size_t r = 0; r < runs.size(); r++) {
                G.runs[4 * r] = runs[r].row + sh; G.runs[4 * r + 1] = runs[r].col + sh;
                G.runs[4 * r + 2] = runs[r].len; G.runs[4 * r + 3] = runs[r].level;
                if (runs[r].level < 0 || runs[r].level >= llcs) failed.store(1);
            }
            if (failed.load()) continue;
            if (!G.dev_built) build_graph(B, store, *mine32, m, n, sh, runs, runs.empty() ? 0 : llcs, G, &jobs);
            if (trace) { ns_runs += (uint64_t)((tb - ta) * 1e6); ns_build += (uint64_t)((now_ms() - tb) * 1e6); }
        }
        } catch (const std::bad_alloc &) { no_memory.store(1); }
    };
    unsigned nt = std::min<unsigned>(16u, std::max(1u, std::thread::hardware_concurrency()));
    {
        // threads pay from a few thousand runs on (a 10 kb tandem-repeat window alone builds 400 000 edges)
        uint64_t load = 0;
        for (const sweep_set &S : sweeps) load += S.flat.size() / 4;
        for (uint64_t i = 0; i < N; i++) load += pair_events[i].n / 2;
        if (N < 64 && load < 4096) nt = 1;
        nt = (unsigned)std::min<uint64_t>(nt, std::max<uint64_t>(N, 1));
    }
    // on the context's kept threads (started one after the other per call, the sixteenth used to begin 0.4 ms after the first)
    malcs_workers *wk = nt > 1 ? malcs_ctx_workers(b->ctx) : nullptr;
    if (wk) nt = std::min<unsigned>(nt, (unsigned)wk->size());
    out->stores.resize(nt);
    out->narrow.resize(nt);
    jobs.builders.store((int)nt);
    if (out->n_dev_built == N) jobs.builders.store(0);                   // every pair's graph came from the device: no pass over the handles
    else if (wk && nt > 1) wk->run((int)nt, [&](int k) { worker(&out->stores[(size_t)k], &out->narrow[(size_t)k]); });
    else worker(&out->stores[0], &out->narrow[0]);
    out->timing[2] = (float)(now_ms() - t0);
    stages.lap("host build");
    if (trace) fprintf(stderr, "malcs host build: %llu pairs, %u threads, wall %.2f ms; summed over threads: events->runs %.2f ms, build_graph %.2f ms\n",
                       (unsigned long long)N, nt, out->timing[2], ns_runs.load() / 1e6, ns_build.load() / 1e6);
    if (no_memory.load()) { malcs_set_error("out of host memory"); return MALCS_ERR_NOMEM; }
    if (failed.load()) { malcs_set_error("device events do not form runs (internal error)"); return MALCS_ERR_INTERNAL; }
    return MALCS_OK;
}

// _lcs_nodes with the reference's complete level lists: distances (family 1), then the sweep kernel
static int nodes_batch(malcs_batch *b, const int64_t *shift, const int64_t *max_distance, malcs_graphs *out)
{
    hipStream_t st = b->ctx->stream;
    const uint64_t N = b->n_pairs;
    out->g.resize(N);
    std::vector<int32_t> dist(N);
    if (N) {
        int rc = malcs_dist_launch(b);
        if (rc) return rc;
        MALCS_HIP(malcs_d2h(st, dist.data(), b->d_dist, N * 4ull));
        MALCS_HIP(malcs_sync(st));
    }
    std::vector<uint32_t> work;
    for (uint64_t i = 0; i < N; i++) {
        pair_graph &G = out->g[i];
        G.distance = dist[i];
        const int64_t md = max_distance ? max_distance[i] : 0;
        if (md && G.distance > md) { G.status = MALCS_PAIR_MAX_DISTANCE_EXCEEDED; continue; }   // :345-346
        if (b->a_len[i] == 0 || b->b_len[i] == 0) continue;                                     // no level lists at all, :319
        work.push_back((uint32_t)i);
    }
    std::vector<uint32_t> flat;
    std::vector<uint64_t> first;
    if (!work.empty()) {
        int rc = malcs_sweep_runs(b, work, dist.data(), flat, first);
        if (rc) return rc;
    }
    std::vector<run> runs;
    for (size_t s = 0; s < work.size(); s++) {
        const uint32_t i = work[s];
        pair_graph &G = out->g[i];
        const int64_t m = b->a_len[i], n = b->b_len[i], sh = shift ? shift[i] : 0;
        runs.clear();
        int64_t top = 0;                                                  // max_lcs_pos, :295,:312
        for (uint64_t r = first[s]; r < first[s + 1]; r++) {
            runs.push_back({flat[4 * r], flat[4 * r + 1], flat[4 * r + 2], flat[4 * r + 3]});
            top = std::max<int64_t>(top, flat[4 * r + 3]);
        }
        order_runs(runs, m, n);
        G.n_levels = top + 1;                                             // lcs_nodes[:max_lcs_pos + 1], :348
        if (out->stores.empty()) out->stores.resize(1);
        G.runs = out->stores[0].take(runs.size() * 4);
        for (size_t r = 0; r < runs.size(); r++) {
            G.runs[4 * r] = runs[r].row + sh; G.runs[4 * r + 1] = runs[r].col + sh;
            G.runs[4 * r + 2] = runs[r].len; G.runs[4 * r + 3] = runs[r].level;
        }
    }
    return MALCS_OK;
}

extern "C" {

int malcs_lcs_nodes_batch(malcs_batch *b, const int64_t *shift, const int64_t *max_distance, malcs_graphs **graphs)
{
    if (!graphs) { malcs_set_error("graphs is NULL"); return MALCS_ERR_ARG; }
    *graphs = nullptr;
    if (!b) { malcs_set_error("batch is NULL"); return MALCS_ERR_ARG; }
    MALCS_HIP(hipSetDevice(b->ctx->device));
    malcs_graphs *g = new (std::nothrow) malcs_graphs();
    if (!g) { malcs_set_error("out of host memory"); return MALCS_ERR_NOMEM; }
    int rc;
    try {
        rc = nodes_batch(b, shift, max_distance, g);
    } catch (const std::bad_alloc &) {
        malcs_set_error("out of host memory");
        rc = MALCS_ERR_NOMEM;
    }
    if (rc) { delete g; return rc; }
    *graphs = g;
    return MALCS_OK;
}

int malcs_graphs_from_runs(uint64_t n_pairs, const int64_t *m, const int64_t *n, const int64_t *shift, const int64_t *n_levels,
                           const int64_t *runs4, const uint64_t *run_first, malcs_graphs **graphs)
{
    if (!graphs) { malcs_set_error("graphs is NULL"); return MALCS_ERR_ARG; }
    *graphs = nullptr;
    if (n_pairs && (!m || !n || !n_levels || !run_first)) { malcs_set_error("NULL argument"); return MALCS_ERR_ARG; }
    malcs_graphs *g = new (std::nothrow) malcs_graphs();
    if (!g) { malcs_set_error("out of host memory"); return MALCS_ERR_NOMEM; }
    int rc = MALCS_OK;
    try {
        g->g.resize(n_pairs);
        g->stores.resize(1);
        g->narrow.resize(1);
        std::unique_ptr<builder> kept = spare_builders().get();
        builder &B = *kept;
        struct give_back { std::unique_ptr<builder> &b; ~give_back() { spare_builders().put(std::move(b)); } } back{kept};
        std::vector<run> runs;
        copy_jobs jobs;                                      // this entry point builds on the calling thread alone
        for (uint64_t i = 0; i < n_pairs && rc == MALCS_OK; i++) {
            pair_graph &G = g->g[i];
            const int64_t sh = shift ? shift[i] : 0;
            if (m[i] < 0 || n[i] < 0 || m[i] > 0x7ffffff0ll || n[i] > 0x7ffffff0ll || run_first[i + 1] < run_first[i]) {
                malcs_set_error("pair %llu: bad lengths or run range", (unsigned long long)i); rc = MALCS_ERR_ARG; break;
            }
            runs.clear();
            int64_t prev = -1;
            for (uint64_t r = run_first[i]; r < run_first[i + 1]; r++) {
                const int64_t *x = runs4 + 4 * r;                        // unshifted row, col; length; level
                if (x[0] < 0 || x[1] < 0 || x[2] < 1 || x[0] + x[2] > m[i] || x[1] + x[2] > n[i] || x[3] < prev || x[3] >= n_levels[i]) {
                    malcs_set_error("pair %llu: run %llu is outside the matrix or the levels are not ascending", (unsigned long long)i,
                                    (unsigned long long)(r - run_first[i]));
                    rc = MALCS_ERR_ARG; break;
                }
                prev = x[3];
                runs.push_back({x[0], x[1], x[2], x[3]});
            }
            if (rc) break;
            G.n_levels = n_levels[i];
            G.distance = m[i] + n[i] - 2 * n_levels[i];                  // len(lcs_nodes) == LLCS whenever there are runs
            G.runs = g->stores[0].take(runs.size() * 4);
            for (size_t r = 0; r < runs.size(); r++) {
                G.runs[4 * r] = runs[r].row + sh; G.runs[4 * r + 1] = runs[r].col + sh; G.runs[4 * r + 2] = runs[r].len; G.runs[4 * r + 3] = runs[r].level;
            }
            build_graph(B, g->stores[0], g->narrow[0], m[i], n[i], sh, runs, runs.empty() ? 0 : n_levels[i], G, &jobs);
        }
    } catch (const std::bad_alloc &) {
        malcs_set_error("out of host memory");
        rc = MALCS_ERR_NOMEM;
    }
    if (rc) { delete g; return rc; }
    *graphs = g;
    return MALCS_OK;
}

int malcs_lcs_graph_batch(malcs_batch *b, const int64_t *shift, const int64_t *max_distance, malcs_graphs **graphs)
{
    if (!graphs) { malcs_set_error("graphs is NULL"); return MALCS_ERR_ARG; }
    *graphs = nullptr;
    if (!b) { malcs_set_error("batch is NULL"); return MALCS_ERR_ARG; }
    MALCS_HIP(hipSetDevice(b->ctx->device));
    malcs_graphs *g = new (std::nothrow) malcs_graphs();
    if (!g) { malcs_set_error("out of host memory"); return MALCS_ERR_NOMEM; }
    int rc;
    try {
        rc = graph_batch(b, shift, max_distance, g);
    } catch (const std::bad_alloc &) {
        malcs_set_error("out of host memory");
        rc = MALCS_ERR_NOMEM;
    }
    if (rc) { delete g; return rc; }
    *graphs = g;
    return MALCS_OK;
}

int malcs_graphs_destroy(malcs_graphs *g) { delete g; return MALCS_OK; }

int malcs_graphs_counts(const malcs_graphs *g, int64_t *counts)
{
    if (!g || !counts) { malcs_set_error("NULL argument"); return MALCS_ERR_ARG; }
    for (size_t i = 0; i < g->g.size(); i++) {
        counts[4 * i] = (int64_t)g->g[i].n_runs();
        counts[4 * i + 1] = g->g[i].n_levels;
        counts[4 * i + 2] = (int64_t)g->g[i].n_nodes();
        counts[4 * i + 3] = (int64_t)g->g[i].n_edges();
    }
    return MALCS_OK;
}

// counts, status, distance and supremal in one pass over the pairs (a handle of 100 000 graphs is 20 MB of per-pair
// records: three passes cost three times the memory traffic)
static void columns_of(const malcs_graphs *g, size_t lo, size_t hi, int64_t *counts, int32_t *status, int64_t *distance, int64_t *supremal4)
{
    for (size_t i = lo; i < hi; i++) {
        const pair_graph &G = g->g[i];
        counts[4 * i] = (int64_t)G.n_runs(); counts[4 * i + 1] = G.n_levels;
        counts[4 * i + 2] = (int64_t)G.n_nodes(); counts[4 * i + 3] = (int64_t)G.n_edges();
        status[i] = G.status;
        distance[i] = G.distance;
        for (int k = 0; k < 4; k++) supremal4[4 * i + k] = G.supremal[k];
    }
}

int malcs_graphs_columns(const malcs_graphs *g, int64_t *counts, int32_t *status, int64_t *distance, int64_t *supremal4)
{
    if (!g || !counts || !status || !distance || !supremal4) { malcs_set_error("NULL argument"); return MALCS_ERR_ARG; }
    columns_of(g, 0, g->g.size(), counts, status, distance, supremal4);
    return MALCS_OK;
}

// malcs_lcs_graph_batch + malcs_graphs_columns in one call: the pass over the handles of a large batch (20 MB per 100 000
// pairs that have just left the cache: 0.9 ms on one thread) is shared among the context's threads
int malcs_lcs_graph_batch_columns(malcs_batch *b, const int64_t *shift, const int64_t *max_distance, malcs_graphs **graphs,
                                  int64_t *counts, int32_t *status, int64_t *distance, int64_t *supremal4)
{
    if (!counts || !status || !distance || !supremal4) { malcs_set_error("NULL argument"); return MALCS_ERR_ARG; }
    const int rc = malcs_lcs_graph_batch(b, shift, max_distance, graphs);
    if (rc) return rc;
    const malcs_graphs *g = *graphs;
    const size_t N = g->g.size();
    malcs_workers *wk = N >= 16384 ? malcs_ctx_workers(b->ctx) : nullptr;
    const int nt = wk ? (int)std::min<size_t>(16, wk->size()) : 1;
    if (nt > 1) wk->run(nt, [&](int t) { columns_of(g, N * (size_t)t / (size_t)nt, N * (size_t)(t + 1) / (size_t)nt, counts, status, distance, supremal4); });
    else columns_of(g, 0, N, counts, status, distance, supremal4);
    return MALCS_OK;
}

int malcs_graphs_status(const malcs_graphs *g, int32_t *status, int64_t *distance)
{
    if (!g) { malcs_set_error("NULL argument"); return MALCS_ERR_ARG; }
    for (size_t i = 0; i < g->g.size(); i++) {
        if (status) status[i] = g->g[i].status;
        if (distance) distance[i] = g->g[i].distance;
    }
    return MALCS_OK;
}

int malcs_graphs_supremal(const malcs_graphs *g, int64_t *supremal4)
{
    if (!g || !supremal4) { malcs_set_error("NULL argument"); return MALCS_ERR_ARG; }
    for (size_t i = 0; i < g->g.size(); i++) memcpy(supremal4 + 4 * i, g->g[i].supremal, sizeof(int64_t) * 4);
    return MALCS_OK;
}

#define MALCS_COPY_OUT(field)                                                                        \
    if (!g || i >= g->g.size() || (!dst && !g->g[i].field.empty())) { malcs_set_error("bad argument"); return MALCS_ERR_ARG; } \
    if (!g->g[i].field.empty()) memcpy(dst, g->g[i].field.data(), g->g[i].field.size() * sizeof(int64_t)); \
    return MALCS_OK;

int malcs_graphs_runs(const malcs_graphs *g, uint64_t i, int64_t *dst)
{
    if (g && i < g->g.size()) widen(const_cast<malcs_graphs *>(g), const_cast<malcs_graphs *>(g)->g[i]);
    MALCS_COPY_OUT(runs)
}
// narrow graphs are widened on the way out (rows, columns and variant positions get their shift), not in place first
int malcs_graphs_nodes(const malcs_graphs *g, uint64_t i, int64_t *dst)
{
    if (g && i < g->g.size() && g->g[i].narrow && !g->g[i].nodes.p) {
        const pair_graph &G = g->g[i];
        if (!dst && G.dev_n_nodes) { malcs_set_error("bad argument"); return MALCS_ERR_ARG; }
        for (int64_t v = 0; v < G.dev_n_nodes; v++) {
            dst[3 * v] = G.node_word(3 * v) + G.dev_shift; dst[3 * v + 1] = G.node_word(3 * v + 1) + G.dev_shift; dst[3 * v + 2] = G.node_word(3 * v + 2);
        }
        return MALCS_OK;
    }
    MALCS_COPY_OUT(nodes)
}
int malcs_graphs_edges(const malcs_graphs *g, uint64_t i, int64_t *dst)
{
    if (g && i < g->g.size() && g->g[i].narrow && !g->g[i].edges.p) {
        const pair_graph &G = g->g[i];
        if (!dst && G.dev_n_edges) { malcs_set_error("bad argument"); return MALCS_ERR_ARG; }
        for (int64_t e = 0; e < G.dev_n_edges; e++) {
            int64_t *o = dst + 6 * e;
            o[0] = G.edge_word(6 * e); o[1] = G.edge_word(6 * e + 1); o[2] = G.edge_word(6 * e + 2) + G.dev_shift;
            o[3] = G.edge_word(6 * e + 3) + G.dev_shift; o[4] = G.edge_word(6 * e + 4); o[5] = G.edge_word(6 * e + 5);
        }
        return MALCS_OK;
    }
    MALCS_COPY_OUT(edges)
}

int malcs_graphs_canonical_count(malcs_graphs *g, uint64_t i, int64_t *count)
{
    if (!g || i >= g->g.size() || !count) { malcs_set_error("bad argument"); return MALCS_ERR_ARG; }
    if (g->g[i].status != MALCS_PAIR_OK) { malcs_set_error("pair %llu has no graph", (unsigned long long)i); return MALCS_ERR_ARG; }
    if (!g->g[i].canon_done) { widen(g, g->g[i]); canonical_of(g->g[i]); }
    *count = (int64_t)g->g[i].canon.size() / 4;
    return MALCS_OK;
}

int malcs_graphs_canonical(malcs_graphs *g, uint64_t i, int64_t *dst)
{
    int64_t n = 0;
    int rc = malcs_graphs_canonical_count(g, i, &n);
    if (rc) return rc;
    if (n && !dst) { malcs_set_error("bad argument"); return MALCS_ERR_ARG; }
    if (n) memcpy(dst, g->g[i].canon.data(), (size_t)n * 4 * sizeof(int64_t));
    return MALCS_OK;
}

int malcs_graphs_atoms(malcs_graphs *g, uint64_t count, const uint64_t *index, const uint8_t *obs, const uint64_t *obs_off,
                       const uint64_t *out_range, uint8_t *out)
{
    if (!g || (count && (!index || !obs_off || !out_range || !out))) { malcs_set_error("NULL argument"); return MALCS_ERR_ARG; }
    std::vector<int32_t> diff;
    for (uint64_t k = 0; k < count; k++) {
        if (index[k] >= g->g.size() || g->g[index[k]].status != MALCS_PAIR_OK) { malcs_set_error("pair %llu has no graph", (unsigned long long)index[k]); return MALCS_ERR_ARG; }
        pair_graph &G = g->g[index[k]];
        widen(g, G);
        const int64_t lo = G.supremal[0], size = G.supremal[1] - lo + 1;
        if (size < 1 || (uint64_t)size != out_range[2 * k + 1] - out_range[2 * k]) {
            malcs_set_error("pair %llu: the supremal spans %lld positions, the caller left room for %llu", (unsigned long long)index[k],
                            (long long)size, (unsigned long long)(out_range[2 * k + 1] - out_range[2 * k]));
            return MALCS_ERR_ARG;
        }
        const uint8_t *o = obs ? obs + obs_off[k] : nullptr;
        const int64_t n_obs = (int64_t)(obs_off[k + 1] - obs_off[k]);
        diff.assign((size_t)(5 * (size + 2)), 0);
        const size_t ne = G.edges.size() / 6;
        for (size_t e = 0; e < ne; e++) {
            const int64_t *x = &G.edges[6 * e];
            const int64_t s = x[2] - lo, t = x[3] - lo;
            if (s < 0 || t < s || t >= size || x[4] < 0 || x[5] < x[4] || x[5] > n_obs) { malcs_set_error("pair %llu: edge outside its supremal or its observed sequence", (unsigned long long)index[k]); return MALCS_ERR_ARG; }
            diff[(size_t)s] += 1; diff[(size_t)t] -= 1;                                         // (idx, "") for idx in [start, end)
            uint32_t present = 0;
            for (int64_t q = x[4]; q < x[5] && present != 15u; q++) {
                if (o[q] > 3) { malcs_set_error("pair %llu: symbol code %u in the observed sequence (atom masks are for codes 0 ... 3)", (unsigned long long)index[k], o[q]); return MALCS_ERR_ARG; }
                present |= 1u << o[q];
            }
            for (int c = 0; c < 4; c++) if (present >> c & 1u) { diff[(size_t)((c + 1) * (size + 2) + s)] += 1; diff[(size_t)((c + 1) * (size + 2) + t + 1)] -= 1; }   // (idx, ch) for idx in [start, end]
        }
        uint8_t *m = out + out_range[2 * k];
        int32_t run[5] = {0, 0, 0, 0, 0};
        for (int64_t q = 0; q < size; q++) {
            uint8_t v = 0;
            for (int c = 0; c < 5; c++) { run[c] += diff[(size_t)(c * (size + 2) + q)]; if (run[c] > 0) v |= (uint8_t)(1u << c); }
            m[q] = v;
        }
    }
    return MALCS_OK;
}

int malcs_graphs_device_built(const malcs_graphs *g, uint64_t *count, float *ms)
{
    if (!g) { malcs_set_error("NULL argument"); return MALCS_ERR_ARG; }
    if (count) *count = g->n_dev_built;
    if (ms) *ms = g->timing[3];
    return MALCS_OK;
}

int malcs_graphs_tie_bytes(const malcs_graphs *g, uint64_t *bytes, uint64_t *pairs)
{
    if (!g) { malcs_set_error("NULL argument"); return MALCS_ERR_ARG; }
    if (bytes) *bytes = g->tie_bytes;
    if (pairs) *pairs = g->tie_pairs;
    return MALCS_OK;
}

int malcs_graphs_timing(const malcs_graphs *g, float *out3)
{
    if (!g || !out3) { malcs_set_error("NULL argument"); return MALCS_ERR_ARG; }
    memcpy(out3, g->timing, 3 * sizeof(float));
    return MALCS_OK;
}

}  // extern "C"
