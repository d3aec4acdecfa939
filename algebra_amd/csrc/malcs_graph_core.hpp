// malcs_graph_core.hpp -- _build_graph on integer ids and flat int32 arrays, one source for the host and the device.
//
// Restates algebra/lcs/lcs_graph.py:351-453 (sink :379-385, level loop with edge creation / node splitting /
// re-insertion :389-426, source :428-445, trimming of source and sink :447-451, supremal :453) for a pair whose
// runs arrive sorted as the reference's level lists.  Coordinates are relative to shift = 0: the
// graph's structure does not depend on the shift, the callers add it to rows, columns and variant positions afterwards.
// Two places of the reference treat the shift unevenly: the degenerate cases :370-377 (they never get here) and the trim
// of the sink :451, whose `max_sink` starts at 0 in SHIFTED coordinates -- a sink no edge ever reaches (identical
// sequences given to LCSgraph() directly) ends up with a length that depends on the shift, so `build` takes it.
//
// Edge lists are persistent: an entry points at the entry appended before it (e_prev) and a node holds its newest
// entry and its edge count, so splitting a node (:405-412, the split takes a copy of the parent's edges and both go on
// growing) is two assignments -- parent and split share every entry that existed at the split, and entries never
// change.  A 10 kb tandem-repeat window makes 400 000 to 900 000 edges out of some 50 000 entries this way; copying
// them at every split was most of the construction's time and all of its memory.
//
// The working state lives in a caller-provided arena (LDS or global memory on the device, vectors on the host).  When
// an array is full the caller's `grow(arena, which)` is asked for more room (the host re-allocates that array and
// re-points the arena; the device has none to give and answers false); without more room `overflow` is set and the
// construction stops without writing out of bounds, so the pair can be handed to the host.  Nothing here touches symbols.
#pragma once
#include <cstdint>

#if defined(__HIPCC__) || defined(__CUDACC__)
#define MALCS_HD __host__ __device__ __forceinline__
#else
#define MALCS_HD inline
#endif

namespace malcs_core {

#ifdef MALCS_CORE_STATS                  // tools/core_bench.cpp: dominance tests, list entries moved, levels with work
extern long long stat_tests, stat_shift, stat_levels;
#define MALCS_STAT(x) (x)
#else
#define MALCS_STAT(x) ((void)0)
#endif

struct arena {
    // nodes (Node.row / .col / .length / ._length / ._incoming, lcs_graph.py:36-44); tail: newest pool entry of the
    // node's out-edges (-1: none), deg: how many it has (Node.edges order = oldest first = the chain backwards)
    int32_t *row, *col, *length, *wlen, *incoming, *deg, *tail;
    int32_t *id, *order;                 // numbering (after the build)
    int32_t node_cap, n_nodes;
    // edge pool, 6 words per entry: child, variant (start, end, observed[from:to]), the entry appended before this one
    int32_t *edge;
    int32_t edge_cap, n_edges;
    // the two level lists that are alive at a time
    int32_t *cur, *below;
    int32_t list_cap, n_cur, n_below;
    int32_t overflow;
};

// int32 words an arena of these capacities needs (nodes: 9 arrays, edges: 6, lists: 2)
MALCS_HD int64_t arena_words(int32_t node_cap, int32_t edge_cap, int32_t list_cap)
{
    return 9ll * node_cap + 6ll * edge_cap + 2ll * list_cap;
}

MALCS_HD void arena_bind(arena &A, int32_t *mem, int32_t node_cap, int32_t edge_cap, int32_t list_cap)
{
    A.row = mem; A.col = A.row + node_cap; A.length = A.col + node_cap; A.wlen = A.length + node_cap;
    A.incoming = A.wlen + node_cap; A.deg = A.incoming + node_cap; A.tail = A.deg + node_cap;
    A.id = A.tail + node_cap; A.order = A.id + node_cap;
    A.edge = A.order + node_cap;
    A.cur = A.edge + 6 * (int64_t)edge_cap; A.below = A.cur + list_cap;
    A.node_cap = node_cap; A.edge_cap = edge_cap; A.list_cap = list_cap;
    A.n_nodes = A.n_edges = A.n_cur = A.n_below = 0;
    A.overflow = 0;
}

enum { E_CHILD = 0, E_START = 1, E_END = 2, E_FROM = 3, E_TO = 4, E_PREV = 5, E_WORDS = 6 };
constexpr int32_t E_COUNTED = (int32_t)0x80000000;        // on E_CHILD: count_reachable has been here
enum { GROW_NODES = 0, GROW_EDGES = 1, GROW_LISTS = 2 };
struct no_growth { MALCS_HD bool operator()(arena &, int) const { return false; } };

template <typename Grow>
MALCS_HD int32_t add_node(arena &A, int32_t r, int32_t c, int32_t l, Grow &grow)
{
    if (A.n_nodes >= A.node_cap && !(grow(A, GROW_NODES) && A.n_nodes < A.node_cap)) { A.overflow = 1; return A.node_cap - 1; }
    const int32_t v = A.n_nodes++;
    A.row[v] = r; A.col[v] = c; A.length[v] = l; A.wlen[v] = l; A.incoming[v] = 0; A.deg[v] = 0; A.tail[v] = -1;
    return v;
}

template <typename Grow>
MALCS_HD void push_edge(arena &A, int32_t node, int32_t child, int32_t start, int32_t end, int32_t from, int32_t to, Grow &grow)
{
    if (A.n_edges >= A.edge_cap && !(grow(A, GROW_EDGES) && A.n_edges < A.edge_cap)) { A.overflow = 1; return; }
    const int32_t slot = A.n_edges++;
    int32_t *e = A.edge + E_WORDS * (int64_t)slot;
    e[E_CHILD] = child; e[E_START] = start; e[E_END] = end; e[E_FROM] = from; e[E_TO] = to; e[E_PREV] = A.tail[node];
    A.tail[node] = slot;
    A.deg[node]++;
}

// runs: 4 int32 per run {row, col, length, level}, sorted by (level, emission order); n_runs >= 1; n_levels = LLCS.
// On return: A holds the graph, source / sink its end nodes, supremal = {start, end, obs_from, obs_to}.
template <typename Grow>
MALCS_HD void build(arena &A, int32_t m, int32_t n, int64_t shift, const int32_t *runs, int32_t n_runs, int32_t n_levels,
                    int32_t &source, int32_t &sink, int32_t (&supremal)[4], Grow grow, bool nodes_filled = false)
{
    // Level lists: the runs arrive sorted by level, so level L is a contiguous range of node ids (node id == run index);
    // only two lists are alive at a time (the level being emptied and the one below it) -- the reference's list of
    // ~LLCS lists would cost an allocation per level.
    // (nodes_filled: the caller has made the nodes of the runs already -- the device's wave does, a lane per run)
    if (nodes_filled && n_runs <= A.node_cap) A.n_nodes = n_runs;
    else for (int32_t r = 0; r < n_runs; r++) add_node(A, runs[4 * r], runs[4 * r + 1], runs[4 * r + 2], grow);
    if (A.overflow || A.list_cap < 1) { A.overflow = 1; return; }         // node id == run index needs every run in the arena
    int32_t unread = n_runs;                                             // runs [0, unread) are not in a list yet
    const int32_t last = n_runs - 1;                                     // last node of the last level, :379-385
    if (A.row[last] + A.length[last] == m && A.col[last] + A.length[last] == n) {
        unread--;                                                        // popped from its level
        sink = add_node(A, A.row[last], A.col[last], A.length[last] + 1, grow);
    } else {
        sink = add_node(A, m, n, 1, grow);
    }
    // the lists are used through the arena's pointers: growing them re-points those
    int32_t n_cur = 0, n_below = 0;
    A.cur[n_cur++] = sink;

    int64_t max_sink = 0;                                                // in shifted coordinates, as the reference's (:387)
    for (int32_t depth = n_levels + 1; depth > 1 && !A.overflow; depth--) {  // depth == len_lcs_nodes, :389
        // Levels without runs of their own under a single live node only shorten that node by one each (no parent to
        // meet, :418-423), under no node at all they do nothing: take them in one stride.  A 1 Mb window a few edits
        // from its reference is almost nothing but such levels.
        const int32_t idle = depth - 2 - (unread ? runs[4 * (unread - 1) + 3] : -1);   // levels until the next run
        if (idle > 0 && n_cur <= 1) {
            int32_t stride = idle;
            if (n_cur == 1) {
                const int32_t node = A.cur[0];
                if (node != sink && A.tail[node] < 0) { n_cur = 0; }
                else { stride = idle < A.wlen[node] - 1 ? idle : A.wlen[node] - 1; A.wlen[node] -= stride; }
            }
            if (stride > 0) { depth -= stride - 1; continue; }
        }
        int32_t first = unread;
        while (first > 0 && runs[4 * (first - 1) + 3] == depth - 2) first--;
        n_below = 0;
        while (unread - first > A.list_cap) if (!grow(A, GROW_LISTS)) { A.overflow = 1; break; }
        if (A.overflow) break;
        for (int32_t i = first; i < unread; i++) A.below[n_below++] = i;
        unread = first;
        MALCS_STAT(stat_levels++);
        for (int32_t qi = 0; qi < n_cur && !A.overflow; qi++) {          // pop(0) in order, :390-391
            const int32_t node = A.cur[qi];
            if (node != sink && A.tail[node] < 0) continue;              // :393-394
            int32_t idx_parent = 0;
            const int32_t nr = A.row[node] + A.wlen[node], nc = A.col[node] + A.wlen[node];
            MALCS_STAT(stat_tests += n_below);
            for (int32_t idx = 0; idx < n_below; idx++) {                // :397
                const int32_t parent = A.below[idx];
                const int32_t pr = A.row[parent] + A.wlen[parent], pc = A.col[parent] + A.wlen[parent];
                if (pr < nr && pc < nc) {                                // :398 strict dominance of the ends
                    if (node == sink && nr - 1 + shift > max_sink) max_sink = nr - 1 + shift;   // :402-403
                    if (A.incoming[parent] == depth) {                   // :405-412 split the parent
                        const int32_t split = add_node(A, A.row[parent], A.col[parent], A.length[parent], grow);
                        A.wlen[split] = A.wlen[parent];
                        A.tail[split] = A.tail[parent];                  // the parent's edges so far, shared
                        A.deg[split] = A.deg[parent];
                        push_edge(A, split, node, pr, nr - 1, pc, nc - 1, grow);               // :399-400
                        A.below[idx] = split;
                        A.row[parent] += A.wlen[parent];
                        A.col[parent] += A.wlen[parent];
                        A.length[parent] -= A.wlen[parent];
                    } else {
                        push_edge(A, parent, node, pr, nr - 1, pc, nc - 1, grow);              // :414
                    }
                    idx_parent = idx + 1;                                // :416
                }
            }
            if (A.wlen[node] > 1) {                                      // :418-423
                A.wlen[node] -= 1;
                if (idx_parent) A.incoming[node] = depth;
                if (n_below >= A.list_cap && !(grow(A, GROW_LISTS) && n_below < A.list_cap)) { A.overflow = 1; break; }
                MALCS_STAT(stat_shift += n_below - idx_parent);
                for (int32_t k = n_below; k > idx_parent; k--) A.below[k] = A.below[k - 1];
                A.below[idx_parent] = node;
                n_below++;
            }
        }
        int32_t *t = A.cur; A.cur = A.below; A.below = t;                // :425
        n_cur = n_below;
    }

    if (A.overflow) return;
    int32_t first = 0;                                                   // level 0, :428-432
    source = n_cur ? A.cur[0] : sink;
    if (n_cur && A.row[source] == 0 && A.col[source] == 0) first = 1;
    else source = add_node(A, 0, 0, 0, grow);
    for (int32_t qi = first; qi < n_cur; qi++) {                         // :434-445
        const int32_t node = A.cur[qi];
        if (node != sink && A.tail[node] < 0) continue;
        const int32_t nr = A.row[node] + A.wlen[node], nc = A.col[node] + A.wlen[node];
        if (A.row[source] < nr && A.col[source] < nc) {
            if (node == sink && nr - 1 + shift > max_sink) max_sink = nr - 1 + shift;
            push_edge(A, source, node, A.row[source], nr - 1, A.col[source], nc - 1, grow);
        }
    }
    int32_t source_offset = 0;                                           // :447-451
    if (A.tail[source] >= 0) {
        source_offset = A.edge[E_WORDS * (int64_t)A.tail[source] + E_START];
        for (int32_t i = A.tail[source]; i >= 0; i = A.edge[E_WORDS * (int64_t)i + E_PREV]) {
            const int32_t start = A.edge[E_WORDS * (int64_t)i + E_START];
            if (start < source_offset) source_offset = start;
        }
    }
    A.row[source] += source_offset;
    A.col[source] += source_offset;
    A.length[source] -= source_offset;
    A.length[sink] -= (int32_t)((int64_t)A.row[sink] + shift + A.length[sink] - max_sink);
    supremal[0] = A.row[source];                                         // :453
    supremal[1] = A.row[sink] + A.length[sink];
    supremal[2] = A.col[source];
    supremal[3] = A.col[sink] + A.length[sink];
    A.n_cur = n_cur; A.n_below = n_below;
}

// The nodes the source reaches and the edges that leave them (any order; the chains are walked newest first, and only
// as far as no walk has been before: what lies behind an entry lay behind it for every node that shares it).  Marks
// the nodes in A.id (-2; others -1) for write_graph and the entries in their child word; uses A.order as the work list.
MALCS_HD int32_t count_reachable(arena &A, int32_t source, int64_t &n_edges)
{
    for (int32_t v = 0; v < A.n_nodes; v++) A.id[v] = -1;
    int32_t count = 0;
    A.id[source] = -2;
    A.order[count++] = source;
    n_edges = 0;
    for (int32_t h = 0; h < count; h++) {
        const int32_t u = A.order[h];
        n_edges += A.deg[u];
        for (int32_t i = A.tail[u]; i >= 0;) {
            int32_t *e = A.edge + E_WORDS * (int64_t)i;
            const int32_t child = e[E_CHILD];
            if (child < 0) break;
            e[E_CHILD] = child | E_COUNTED;
            if (A.id[child] == -1) { A.id[child] = -2; A.order[count++] = child; }
            i = e[E_PREV];
        }
    }
    return count;
}

// Canonical numbering (BFS by identity from the source, following edge order) and the arrays in one pass, after
// count_reachable.  nodes: 3 per node {row, col, length}; edges: 6 per edge {src, dst, start, end, obs_from, obs_to},
// grouped by src in node order and, within a node, in the order of Node.edges.  A node's chain runs from its newest edge
// to its oldest; a child is numbered when an edge first leads to it, in edge order.
// `where` (one word per pool entry; null: see below): the place in `edges` where an entry has been written.  A node's
// edges are written from the back of their range walking the chain, then numbered walking the range forwards.  The
// chain behind an entry is the same for every node that shares it, so a walk that meets a written entry copies that
// node's edges up to it (children numbered already, only the source differs) instead of walking on: every entry is
// walked once, the rest is a block copy -- a tandem-repeat graph's 600 000 edges are 50 000 entries.
template <typename T>
MALCS_HD void write_graph(arena &A, int32_t source, T *nodes, T *edges, int64_t shift, int32_t *where = nullptr,
                          int32_t *n_nodes_out = nullptr, int64_t *n_edges_out = nullptr)
{
    if (where) for (int32_t i = 0; i < A.n_edges; i++) where[i] = -1;
    int32_t seen = 0;
    A.id[source] = 0;
    A.order[seen++] = source;
    int64_t ei = 0;
    for (int32_t h = 0; h < seen; h++) {
        const int32_t u = A.order[h], d = A.deg[u];
        nodes[3 * h] = (T)(A.row[u] + shift); nodes[3 * h + 1] = (T)(A.col[u] + shift); nodes[3 * h + 2] = (T)A.length[u];
        if (!where) {
            // No table: turn the chain round in place (oldest first), write and number along it, and turn every link
            // back on the way -- two walks through the arena, nothing read back from `edges` (device memory there).
            int32_t fwd = -1;
            for (int32_t i = A.tail[u]; i >= 0;) {
                int32_t *e = A.edge + E_WORDS * (int64_t)i;
                const int32_t older = e[E_PREV];
                e[E_PREV] = fwd; fwd = i; i = older;
            }
            T *o = edges + 6 * ei;
            int32_t back = -1;
            for (int32_t i = fwd; i >= 0; o += 6) {
                int32_t *e = A.edge + E_WORDS * (int64_t)i;
                const int32_t child = e[E_CHILD] & ~E_COUNTED;
                if (A.id[child] < 0) { A.id[child] = seen; A.order[seen++] = child; }
                o[0] = (T)h; o[1] = (T)A.id[child]; o[2] = (T)(e[E_START] + shift); o[3] = (T)(e[E_END] + shift);
                o[4] = (T)e[E_FROM]; o[5] = (T)e[E_TO];
                const int32_t newer = e[E_PREV];
                e[E_PREV] = back; back = i; i = newer;
            }
            ei += d;
            continue;
        }
        int32_t k = d, i = A.tail[u];                        // edges [k, d) of the node have been written from the chain
        while (i >= 0 && where[i] < 0) {
            const int32_t *e = A.edge + E_WORDS * (int64_t)i;
            k--;
            T *o = edges + 6 * (ei + k);
            o[0] = (T)h; o[1] = (T)(e[E_CHILD] & ~E_COUNTED); o[2] = (T)(e[E_START] + shift); o[3] = (T)(e[E_END] + shift);
            o[4] = (T)e[E_FROM]; o[5] = (T)e[E_TO];
            if (ei + k <= 0x7fffffff) where[i] = (int32_t)(ei + k);
            i = e[E_PREV];
        }
        if (i >= 0) {                                        // a written entry: it is edge k - 1 here and there
            const T *from = edges + 6 * ((int64_t)where[i] - (k - 1));
            T *to = edges + 6 * ei;
            __builtin_memcpy(to, from, sizeof(T) * 6 * (uint64_t)k);         // an earlier node's range: no overlap
            for (int32_t s = 0; s < k; s++) to[6 * s] = (T)h;
        }
        T *o = edges + 6 * ei;
        for (int32_t s = k; s < d; s++) {
            const int32_t child = (int32_t)o[6 * s + 1];
            if (A.id[child] < 0) { A.id[child] = seen; A.order[seen++] = child; }
            o[6 * s + 1] = (T)A.id[child];
        }
        ei += d;
    }
    if (n_nodes_out) *n_nodes_out = seen;                    // (a caller that has not counted beforehand: it needs A.id = -1 for every node)
    if (n_edges_out) *n_edges_out = ei;
}

// write_graph in two parts, for graphs large enough to share the copying among threads (host only; same arrays).
// Part 1 (one thread, in node order): the nodes, every edge a node is the first to walk (numbered), and per node h where
// its edges start (at[h]), how many of them are another node's first edges (pre[h]) and whose (from[h], a node number
// below h; -1: none).  `where` here holds the NUMBER of the node that first walked an entry.
template <typename T>
MALCS_HD void write_graph_own(arena &A, int32_t source, T *nodes, T *edges, int64_t shift, int32_t *where, int32_t *at, int32_t *pre,
                              int32_t *from)
{
    for (int32_t i = 0; i < A.n_edges; i++) where[i] = -1;
    int32_t seen = 0;
    A.id[source] = 0;
    A.order[seen++] = source;
    int64_t ei = 0;
    for (int32_t h = 0; h < seen; h++) {
        const int32_t u = A.order[h], d = A.deg[u];
        nodes[3 * h] = (T)(A.row[u] + shift); nodes[3 * h + 1] = (T)(A.col[u] + shift); nodes[3 * h + 2] = (T)A.length[u];
        at[h] = (int32_t)ei;
        int32_t k = d, i = A.tail[u];
        T *o = edges + 6 * ei;
        while (i >= 0 && where[i] < 0) {
            const int32_t *e = A.edge + E_WORDS * (int64_t)i;
            k--;
            const int32_t child = e[E_CHILD] & ~E_COUNTED;
            T *w = o + 6 * k;
            w[0] = (T)h; w[1] = (T)child; w[2] = (T)(e[E_START] + shift); w[3] = (T)(e[E_END] + shift);
            w[4] = (T)e[E_FROM]; w[5] = (T)e[E_TO];
            where[i] = h;
            i = e[E_PREV];
        }
        pre[h] = i >= 0 ? k : 0;
        from[h] = i >= 0 ? where[i] : -1;
        for (int32_t s = k; s < d; s++) {                    // the children of the first k edges were numbered at from[h]
            const int32_t child = (int32_t)o[6 * s + 1];
            if (A.id[child] < 0) { A.id[child] = seen; A.order[seen++] = child; }
            o[6 * s + 1] = (T)A.id[child];
        }
        ei += d;
    }
}

// Part 2 (any thread, any nodes [lo, hi), after part 1): the first pre[h] edges of node h are the first pre[h] edges of
// from[h], of which the first pre[from[h]] are somebody else's in turn: every stretch is copied from the node that
// wrote it itself, so no copy waits for another.  (memcpy and a second pass for the source column: a single pass that
// writes six words per edge measured half as fast.)
template <typename T>
inline void copy_prefixes(T *edges, const int32_t *at, const int32_t *pre, const int32_t *from, int32_t lo, int32_t hi)
{
    for (int32_t h = lo; h < hi; h++) {
        int32_t want = pre[h], g = from[h];
        T *dst = edges + 6 * (int64_t)at[h];
        while (want > 0) {
            const int32_t pg = pre[g];
            if (want > pg) {
                __builtin_memcpy(dst + 6 * (int64_t)pg, edges + 6 * ((int64_t)at[g] + pg), sizeof(T) * 6 * (uint64_t)(want - pg));
                for (int32_t s = pg; s < want; s++) dst[6 * (int64_t)s] = (T)h;
                want = pg;
            }
            g = from[g];
        }
    }
}

}  // namespace malcs_core
