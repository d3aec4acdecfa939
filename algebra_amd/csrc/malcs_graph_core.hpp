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

struct arena {
    // nodes (Node.row / .col / .length / ._length / ._incoming, lcs_graph.py:36-44); head / tail: first / last pool
    // entry of the node's out-edges (append order = Node.edges order), -1: none
    int32_t *row, *col, *length, *wlen, *incoming, *head, *tail;
    int32_t *id, *order;                 // numbering (after the build)
    int32_t node_cap, n_nodes;
    // edge pool: child, variant (start, end, observed[from:to]), next entry of the same node
    int32_t *e_child, *e_start, *e_end, *e_from, *e_to, *e_next;
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
    A.incoming = A.wlen + node_cap; A.head = A.incoming + node_cap; A.tail = A.head + node_cap;
    A.id = A.tail + node_cap; A.order = A.id + node_cap;
    A.e_child = A.order + node_cap; A.e_start = A.e_child + edge_cap; A.e_end = A.e_start + edge_cap;
    A.e_from = A.e_end + edge_cap; A.e_to = A.e_from + edge_cap; A.e_next = A.e_to + edge_cap;
    A.cur = A.e_next + edge_cap; A.below = A.cur + list_cap;
    A.node_cap = node_cap; A.edge_cap = edge_cap; A.list_cap = list_cap;
    A.n_nodes = A.n_edges = A.n_cur = A.n_below = 0;
    A.overflow = 0;
}

enum { GROW_NODES = 0, GROW_EDGES = 1, GROW_LISTS = 2 };
struct no_growth { MALCS_HD bool operator()(arena &, int) const { return false; } };

template <typename Grow>
MALCS_HD int32_t add_node(arena &A, int32_t r, int32_t c, int32_t l, Grow &grow)
{
    if (A.n_nodes >= A.node_cap && !(grow(A, GROW_NODES) && A.n_nodes < A.node_cap)) { A.overflow = 1; return A.node_cap - 1; }
    const int32_t v = A.n_nodes++;
    A.row[v] = r; A.col[v] = c; A.length[v] = l; A.wlen[v] = l; A.incoming[v] = 0; A.head[v] = -1; A.tail[v] = -1;
    return v;
}

template <typename Grow>
MALCS_HD void push_edge(arena &A, int32_t node, int32_t child, int32_t start, int32_t end, int32_t from, int32_t to, Grow &grow)
{
    if (A.n_edges >= A.edge_cap && !(grow(A, GROW_EDGES) && A.n_edges < A.edge_cap)) { A.overflow = 1; return; }
    const int32_t slot = A.n_edges++;
    A.e_child[slot] = child; A.e_start[slot] = start; A.e_end[slot] = end; A.e_from[slot] = from; A.e_to[slot] = to;
    A.e_next[slot] = -1;
    if (A.tail[node] < 0) A.head[node] = slot; else A.e_next[A.tail[node]] = slot;
    A.tail[node] = slot;
}

template <typename Grow>
MALCS_HD void copy_edges(arena &A, int32_t dst, int32_t src, Grow &grow)
{
    for (int32_t i = A.head[src]; i >= 0; i = A.e_next[i]) {
        if (A.overflow) return;
        push_edge(A, dst, A.e_child[i], A.e_start[i], A.e_end[i], A.e_from[i], A.e_to[i], grow);
    }
}

// runs: 4 int32 per run {row, col, length, level}, sorted by (level, emission order); n_runs >= 1; n_levels = LLCS.
// On return: A holds the graph, source / sink its end nodes, supremal = {start, end, obs_from, obs_to}.
template <typename Grow>
MALCS_HD void build(arena &A, int32_t m, int32_t n, int64_t shift, const int32_t *runs, int32_t n_runs, int32_t n_levels,
                    int32_t &source, int32_t &sink, int32_t (&supremal)[4], Grow grow)
{
    // Level lists: the runs arrive sorted by level, so level L is a contiguous range of node ids (node id == run index);
    // only two lists are alive at a time (the level being emptied and the one below it) -- the reference's list of
    // ~LLCS lists would cost an allocation per level.
    for (int32_t r = 0; r < n_runs; r++) add_node(A, runs[4 * r], runs[4 * r + 1], runs[4 * r + 2], grow);
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
                if (node != sink && A.head[node] < 0) { n_cur = 0; }
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
        for (int32_t qi = 0; qi < n_cur && !A.overflow; qi++) {          // pop(0) in order, :390-391
            const int32_t node = A.cur[qi];
            if (node != sink && A.head[node] < 0) continue;              // :393-394
            int32_t idx_parent = 0;
            const int32_t nr = A.row[node] + A.wlen[node], nc = A.col[node] + A.wlen[node];
            for (int32_t idx = 0; idx < n_below; idx++) {                // :397
                const int32_t parent = A.below[idx];
                const int32_t pr = A.row[parent] + A.wlen[parent], pc = A.col[parent] + A.wlen[parent];
                if (pr < nr && pc < nc) {                                // :398 strict dominance of the ends
                    if (node == sink && nr - 1 + shift > max_sink) max_sink = nr - 1 + shift;   // :402-403
                    if (A.incoming[parent] == depth) {                   // :405-412 split the parent
                        const int32_t split = add_node(A, A.row[parent], A.col[parent], A.length[parent], grow);
                        A.wlen[split] = A.wlen[parent];
                        copy_edges(A, split, parent, grow);
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
        if (node != sink && A.head[node] < 0) continue;
        const int32_t nr = A.row[node] + A.wlen[node], nc = A.col[node] + A.wlen[node];
        if (A.row[source] < nr && A.col[source] < nc) {
            if (node == sink && nr - 1 + shift > max_sink) max_sink = nr - 1 + shift;
            push_edge(A, source, node, A.row[source], nr - 1, A.col[source], nc - 1, grow);
        }
    }
    int32_t source_offset = 0;                                           // :447-451
    if (A.head[source] >= 0) {
        source_offset = A.e_start[A.head[source]];
        for (int32_t i = A.head[source]; i >= 0; i = A.e_next[i]) if (A.e_start[i] < source_offset) source_offset = A.e_start[i];
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

// Canonical numbering: BFS by identity from the source, following edge order.  Fills A.id / A.order; returns the number
// of nodes reached and, in n_edges, the number of edges leaving them.
MALCS_HD int32_t number_nodes(arena &A, int32_t source, int32_t &n_edges)
{
    for (int32_t v = 0; v < A.n_nodes; v++) A.id[v] = -1;
    int32_t count = 0;
    A.id[source] = 0;
    A.order[count++] = source;
    n_edges = 0;
    for (int32_t h = 0; h < count; h++) {
        for (int32_t i = A.head[A.order[h]]; i >= 0; i = A.e_next[i]) {
            const int32_t child = A.e_child[i];
            if (A.id[child] < 0) { A.id[child] = count; A.order[count++] = child; }
            n_edges++;
        }
    }
    return count;
}

// nodes: 3 per node {row, col, length}; edges: 6 per edge {src, dst, start, end, obs_from, obs_to}, grouped by src in
// node order and, within a node, in the order of Node.edges
template <typename T>
MALCS_HD void write_graph(const arena &A, int32_t count, T *nodes, T *edges, int64_t shift)
{
    int32_t ei = 0;
    for (int32_t h = 0; h < count; h++) {
        const int32_t u = A.order[h];
        nodes[3 * h] = (T)(A.row[u] + shift); nodes[3 * h + 1] = (T)(A.col[u] + shift); nodes[3 * h + 2] = (T)A.length[u];
        for (int32_t i = A.head[u]; i >= 0; i = A.e_next[i]) {
            T *o = edges + 6 * (int64_t)ei++;
            o[0] = (T)h; o[1] = (T)A.id[A.e_child[i]]; o[2] = (T)(A.e_start[i] + shift); o[3] = (T)(A.e_end[i] + shift);
            o[4] = (T)A.e_from[i]; o[5] = (T)A.e_to[i];
        }
    }
}

}  // namespace malcs_core
