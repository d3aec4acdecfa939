/*
 * malcs.h -- C-ABI of libmalcs.so: the MI355X (gfx950) backend for the LCS hot path of
 * mutalyzer/algebra (algebra/lcs: edit_distance, LCSgraph).
 *
 * The reference has no FFI layer: its boundary for this path is the Python module API of
 * `algebra.lcs` (algebra/lcs/__init__.py:4-11).  Every entry point below names the reference
 * function it stands in for (paths relative to the reference checkout).  The Python facade
 * `algebra_amd.lcs` binds these symbols with ctypes and keeps the reference's names, argument
 * meaning and exceptions; INTEGRATION.md shows the stub a maintainer adds to algebra/lcs.
 *
 * Conventions
 *   - plain pointers and sizes only; all buffers are caller-allocated and caller-owned unless a
 *     handle type says otherwise; handles are freed with their *_destroy function.
 *   - every function returns MALCS_OK (0) or a negative error code; malcs_last_error() returns a
 *     thread-local human-readable message for the last failure on the calling thread.
 *   - symbols are opaque uint8 codes compared for equality only (the reference compares str code
 *     points with ==, algebra/lcs/edit_distance.py:26); `n_symbols` is an exclusive upper bound
 *     on the codes in a batch (DNA: 4) and selects the number of bit planes on the device.
 *   - there is no CPU fallback: without a usable HIP device every compute entry point fails
 *     with MALCS_ERR_HIP.
 *   - one process per GPU: a context is bound to one device; calls on one context are
 *     serialised by the caller.
 */
#ifndef MALCS_H
#define MALCS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MALCS_VERSION 100           /* 0.1.0 */

#define MALCS_OK 0
#define MALCS_ERR_ARG (-1)          /* bad argument */
#define MALCS_ERR_NOMEM (-2)        /* host or device allocation failed */
#define MALCS_ERR_HIP (-3)          /* HIP runtime error / no device */
#define MALCS_ERR_TOO_LONG (-4)     /* a sequence exceeds what the kernels support */
#define MALCS_ERR_INTERNAL (-5)

/* per-pair status values written by malcs_lcs_graph_batch */
#define MALCS_PAIR_OK 0
#define MALCS_PAIR_MAX_DISTANCE_EXCEEDED 1   /* -> ValueError("maximum distance exceeded"), lcs_graph.py:345-346 */

typedef struct malcs_ctx malcs_ctx;
typedef struct malcs_batch malcs_batch;
typedef struct malcs_graphs malcs_graphs;

/* ---- library / context ----------------------------------------------------------------- */

int malcs_version(void);
const char *malcs_last_error(void);
int malcs_device_count(int *count);

/* Bind a context (stream, scratch) to HIP device `device`.  Contexts share nothing but the device: a caller with a
 * stream of batches makes two or three on the same device and puts consecutive batches on them in turn, so that one
 * batch's plan (launches that mostly wait for memory) runs beside another's distance kernels (bound by the VALU);
 * more than the card's four hardware queues gains nothing.  A context is not thread-safe; different contexts may be
 * used from different threads. */
int malcs_ctx_create(int device, malcs_ctx **ctx);
int malcs_ctx_destroy(malcs_ctx *ctx);
int malcs_ctx_sync(malcs_ctx *ctx);
/* Device name / CU count / clock (MHz), for bench reports. */
int malcs_ctx_device_info(malcs_ctx *ctx, char *name, int name_cap, int *cus, int *clock_mhz);

/* ---- batches: sequence pairs resident in HBM ---------------------------------------------
 * `syms` holds all sequences back to back; pair i is (syms[a_off[i] .. +a_len[i]),
 * syms[b_off[i] .. +b_len[i])) = (reference window, observed window).  The call copies the
 * bytes to the device, packs them into bit planes there and returns; the host buffers may be
 * reused immediately.  Pairs may share sequences (same offsets). */
int malcs_batch_create(malcs_ctx *ctx, const uint8_t *syms, uint64_t n_syms,
                       const uint64_t *a_off, const uint32_t *a_len,
                       const uint64_t *b_off, const uint32_t *b_len,
                       uint64_t n_pairs, int n_symbols, malcs_batch **batch);
/* The same batch from sequences that are ALREADY PACKED (a caller that keeps its reference / observed windows as bit
 * planes -- on the host or, planes_on_device != 0, in HBM on the context's device -- skips the pack of the bytes; the
 * reference has no such stage: edit_distance(a, b) takes str, algebra/lcs/edit_distance.py:21).
 *
 * The arena, `planes[n_dwords]`: a sequence of L symbols is a run of ceil(L / 32) BLOCKS; a block is `n_planes`
 * consecutive dwords; bit j (least significant first) of dword p of block k is bit p of the code of symbol 32 k + j.
 * n_planes = 2 for up to 4 symbols (DNA), 3 for up to 8, 4 for up to 16, 8 for up to 256.  Bits behind a sequence's last
 * symbol may hold anything (the library clears them in its copy).  a_off[i] / b_off[i] = the dword at which the pair's
 * reference / observed sequence begins (a multiple of n_planes; its whole run lies inside the arena); a_len / b_len in
 * symbols.  Pairs may share sequences; an empty sequence may point anywhere inside the arena.  The four tables are host
 * arrays.  The call copies the arena (it is the caller's again on return: a device-side arena must not be written by
 * other streams before the context's stream has passed the copy -- malcs_ctx_sync) and plans the batch as
 * malcs_batch_create does.  Errors: MALCS_ERR_ARG (n_planes, a run outside the arena or not block-aligned),
 * MALCS_ERR_TOO_LONG (an arena of 2^32 - 1 dwords or more). */
int malcs_batch_create_packed(malcs_ctx *ctx, const uint32_t *planes, uint64_t n_dwords, int planes_on_device,
                              const uint32_t *a_off, const uint32_t *a_len,
                              const uint32_t *b_off, const uint32_t *b_len,
                              uint64_t n_pairs, int n_planes, malcs_batch **batch);
int malcs_batch_destroy(malcs_batch *batch);
uint64_t malcs_batch_size(const malcs_batch *batch);
/* Queues the part of a batch's distance plan that depends on what its sequences hold, again, on the planes that are
 * already resident: for the window class the common ends of every pair, the sort into tiles, the mode of the class
 * (from a sample), the pieces and their deal to the SIMDs, the rows of every piece -- all of it kernels on the
 * context's stream, nothing is copied back and the call does not wait.  (What follows from the LENGTHS alone -- which
 * pair runs in which class -- is worked out once, on the host, by malcs_batch_create.)  The results of the batch do not
 * change; bench.py times plan + launch with it: what a caller pays who computes a batch's distances once. */
int malcs_batch_replan(malcs_batch *batch);

/* ---- kernel family 1: batched distances ---------------------------------------------------
 * Stands in for edit_distance(a, b) (algebra/lcs/edit_distance.py:21-53):
 * dist[i] = a_len[i] + b_len[i] - 2 * LLCS(a_i, b_i).
 *
 * Three kernels share the work, chosen per pair when the batch is created: similar pairs of some
 * length first run in a window of a few limbs around the main diagonal (exact when the result is at
 * most the window's threshold, otherwise redone), everything else over the full matrix, sequences
 * beyond a resident pattern in a sliding band.  The results are the same whichever kernel ran.
 *
 * _launch enqueues the kernels on the context's stream and returns; the distances stay in HBM.
 * _fetch waits for them and copies them to `dist` (n_pairs int32).  _time runs `iters`
 * back-to-back launches between two HIP events on that stream and reports the elapsed
 * milliseconds (inputs and outputs device-resident; this is what bench.py times). */
int malcs_edit_distance_launch(malcs_batch *batch);
int malcs_edit_distance_fetch(malcs_batch *batch, int32_t *dist);
int malcs_edit_distance_time(malcs_batch *batch, int iters, float *ms_total);

/* What the planner decided for this batch (for reports; bench.py prices its roofline with it).  The plan of the window
 * class lives in device memory; this call waits for the context's stream and copies its summary back.
 * info[0] pairs in the window class, [1] limbs (32 rows each) in the window, [2] its threshold T
 * (results <= T are final, others are redone over the full matrix; 48 when every tile runs in two-limb mode), [3] cells
 * the window class computes, as laid out: rows of the tile's mode x the text of the tile's longest pair, grown to whole
 * blocks of 32 columns, [4] full-matrix cells of the whole batch = sum m * n,
 * [5] number of full-matrix launch classes, [6] tiles (64 pairs) of the window class, [7] of those, tiles in two-limb
 * mode (64 rows, threshold 48: chosen when a sample of the batch makes a pair beyond it improbable),
 * [8] pieces (workgroups that have work), [9] tiles that run as two pieces, [10] workgroups of a launch (an upper bound of
 * [8], known without asking the device), [11] the cells between the common ends of every pair as they are (rows of the
 * mode x text columns: no growth to blocks or to the tile's longest), [12], [13] mean and standard deviation of the sample's
 * distance estimates x 16, [14] what the fullest SIMD of a launch holds (quarters of a three-limb block), [15] whether
 * the cuts were balanced between the SIMDs that hold one piece more and the others. */
int malcs_batch_plan_info(malcs_batch *batch, uint64_t *info16);
/* For tests and reports: copies one of the arrays of the window class's plan from device memory (waits for the context's
 * stream).  what: 0 per-pair records in pair order (16 dwords each: pattern / text offset, rows, columns, ... blocks), 1 the
 * same in slot order (tile = 64 consecutive slots), 2 / 3 first row and blocks | flags of every place of the layout,
 * 4 - 7 first row, blocks | flags, join slot and place of the piece every workgroup of a launch runs, 8 the plan's header
 * and histogram scans.  *n_dwords = size of the array; at most `cap` dwords are copied.  No counterpart in the reference. */
int malcs_batch_plan_dump(malcs_batch *batch, int what, uint32_t *out, uint64_t cap, uint64_t *n_dwords);
/* As _time, but only the batch's dominant kernel (the window kernel when the batch has a window class,
 * else the full-matrix launches), so that its duration can be set against rocprofv3's. */
int malcs_edit_distance_time_kernel(malcs_batch *batch, int iters, float *ms_total);
/* The window kernel of the batch launched `iters` times back to back, every launch between a HIP event pair of its own
 * (hipExtLaunchKernelGGL's start / stop events: the dispatch's time stamps, i.e. what rocprofv3 lists as the kernel's
 * duration; the gap between two kernels of a stream is in _time_kernel's total and not in these).  ms_each[iters].
 * Measurement only, like the two above: no counterpart in the reference.  Error when the batch has no window class. */
int malcs_edit_distance_kernel_durations(malcs_batch *batch, int iters, float *ms_each);
/* As _kernel_durations, but every launch is the window kernel of a whole step (malcs_batch_replan +
 * malcs_edit_distance_launch, `iters` steps queued back to back): the kernel as it runs behind the plan that has just
 * written its rows, which is a few microseconds slower than a replay.  Measurement only; no counterpart in the reference. */
int malcs_edit_distance_step_durations(malcs_batch *batch, int iters, float *ms_each);

/* One-shot convenience: create + launch + fetch + destroy. */
int malcs_edit_distance_batch(malcs_ctx *ctx, const uint8_t *syms, uint64_t n_syms,
                              const uint64_t *a_off, const uint32_t *a_len,
                              const uint64_t *b_off, const uint32_t *b_len,
                              uint64_t n_pairs, int n_symbols, int32_t *dist);

/* ---- the entry point that allocates nothing ---------------------------------------------------
 * NO COUNTERPART IN THE REFERENCE.  For callers that cannot allocate device memory: code under HIP graph
 * capture, tight serving loops, harnesses that forbid hipMalloc.  Same result as malcs_edit_distance_batch
 * (edit_distance(a, b), algebra/lcs/edit_distance.py:21-53), but inputs and output are DEVICE addresses on the
 * context's device and the call performs no hipMalloc / hipFree / host-device copy / synchronisation: it enqueues
 * its kernels on `stream` (a hipStream_t; NULL = the context's stream) and returns.  Read d_dist after that
 * stream has been synchronised (malcs_ctx_sync for the context's stream).
 *
 * Everything between input and output (bit planes, per-pair tables, work lists, counters) lives in one
 * file-scope device array of the library (14 MB per GPU, held until the process ends).  Limits, checked here:
 * n_pairs <= MALCS_NOALLOC_MAX_PAIRS, n_syms <= MALCS_NOALLOC_MAX_SYMS, 1 <= n_symbols <= 16 (else
 * MALCS_ERR_TOO_LONG / MALCS_ERR_ARG, nothing is enqueued).  Checked on the device, per pair (d_dist[i] = -1,
 * the other pairs are unaffected): a sequence outside [0, n_syms), a sequence that holds a symbol code that is not
 * below n_symbols (malcs_batch_create rejects such a batch as a whole), a shorter sequence of more than
 * MALCS_NOALLOC_MAX_LEN symbols, or packed sequences beyond the arena (the sum over all 2 n_pairs sequences of
 * ceil(len / 32) must not exceed MALCS_NOALLOC_PLANE_DWORDS / planes, planes = 2, 3, 4 for <= 4, 8, 16 symbols).
 * All calls on one GPU share that array: calls that overlap on one GPU (two streams, two threads) race. */
#define MALCS_NOALLOC_MAX_PAIRS 65536u
#define MALCS_NOALLOC_MAX_LEN 16384u
#define MALCS_NOALLOC_MAX_SYMS (32ull << 20)
#define MALCS_NOALLOC_PLANE_DWORDS (2u << 20)
int malcs_edit_distance_noalloc(malcs_ctx *ctx, void *stream, const uint8_t *d_syms, uint64_t n_syms,
                                const uint64_t *d_a_off, const uint32_t *d_a_len,
                                const uint64_t *d_b_off, const uint32_t *d_b_len,
                                uint64_t n_pairs, int n_symbols, int32_t *d_dist);

/* ---- kernel family 2: batched LCS graphs ----------------------------------------------------
 * Stands in for LCSgraph.__init__(reference, observed, shift, max_distance)
 * (algebra/lcs/lcs_graph.py:59-61) = _lcs_nodes (:238-348) + _build_graph (:351-453), for every
 * pair of the batch.  `shift` and `max_distance` may be NULL (all 0 / no limit; a max_distance
 * of 0 means no limit, as in the reference, :345).
 *
 * The device computes the distance and the optimal maximal match runs (row, col, length, level),
 * orders them as the reference's sweep does and compresses them into the graph exactly as _build_graph
 * does (kernel family 2c: one source, malcs_graph_core.hpp, compiled for the device and for the host;
 * the host half builds what the device builder does not take: graphs of the bit-vector paths and graphs
 * beyond its run limit).  Results are held by the returned handle.
 *
 * Per pair i the handle exposes
 *   status     MALCS_PAIR_*                      distance   graph.distance
 *   supremal   {start, end, obs_from, obs_to}    graph.supremal = Variant(start, end, observed[obs_from:obs_to])
 *   runs       n_runs x {row, col, length, level}, sorted as the reference's level lists
 *              (shifted coordinates, as LCSgraph.Node)
 *   nodes      n_nodes x {row, col, length}      node 0 is graph._source; numbering = BFS by
 *                                                identity from the source following edge order
 *   edges      n_edges x {src, dst, start, end, obs_from, obs_to}, grouped by src in node order and,
 *              within a node, in the order of Node.edges; the edge's variant is
 *              Variant(start, end, observed[obs_from:obs_to])
 */
int malcs_lcs_graph_batch(malcs_batch *batch, const int64_t *shift, const int64_t *max_distance,
                          malcs_graphs **graphs);
int malcs_graphs_destroy(malcs_graphs *graphs);
/* Stands in for _build_graph(reference, observed, lcs_nodes, shift) (algebra/lcs/lcs_graph.py:351-453) called on its
 * own, on level lists the caller already has (the reference's tests do that, tests/lcs/test_lcs_graph.py): host only,
 * needs no device.  Pair i has sequences of m[i] / n[i] symbols and the runs run_first[i] .. run_first[i + 1] of
 * runs4 = {row, col, length, level} (UNSHIFTED coordinates, in the order of the level lists, levels ascending);
 * n_levels[i] = len(lcs_nodes).  The handle holds nodes, edges, supremal as malcs_lcs_graph_batch's does
 * (distance = m + n - 2 * n_levels). */
int malcs_graphs_from_runs(uint64_t n_pairs, const int64_t *m, const int64_t *n, const int64_t *shift,
                           const int64_t *n_levels, const int64_t *runs4, const uint64_t *run_first,
                           malcs_graphs **graphs);
/* Stands in for _lcs_nodes(reference, observed, shift, max_distance) (algebra/lcs/lcs_graph.py:238-348) called on
 * its own: the distance and the level lists exactly as the reference's sweep fills them, i.e. with the runs it
 * walks over that lie on no minimal alignment as well (its tests pin them, tests/lcs/test_lcs_graph.py:20-51;
 * _build_graph discards them, which is why malcs_lcs_graph_batch does not compute them).  The returned handle
 * holds status, distance, n_levels (= len of the reference's list) and runs only; nodes and edges are empty. */
int malcs_lcs_nodes_batch(malcs_batch *batch, const int64_t *shift, const int64_t *max_distance,
                          malcs_graphs **graphs);
/* per-pair sizes: counts[4*i + {0,1,2,3}] = {n_runs, n_levels, n_nodes, n_edges} */
int malcs_graphs_counts(const malcs_graphs *graphs, int64_t *counts);
int malcs_graphs_status(const malcs_graphs *graphs, int32_t *status, int64_t *distance);
int malcs_graphs_supremal(const malcs_graphs *graphs, int64_t *supremal4);
/* the three calls above in one pass over the pairs (what the facade reads out of every handle) */
int malcs_graphs_columns(const malcs_graphs *graphs, int64_t *counts, int32_t *status, int64_t *distance, int64_t *supremal4);
/* malcs_lcs_graph_batch and malcs_graphs_columns in one call (same arguments, same results): the pass over the pairs of
 * a large batch runs on the context's threads instead of the caller's one. */
int malcs_lcs_graph_batch_columns(malcs_batch *batch, const int64_t *shift, const int64_t *max_distance, malcs_graphs **graphs,
                                  int64_t *counts, int32_t *status, int64_t *distance, int64_t *supremal4);
/* fill caller buffers for pair i (sizes from malcs_graphs_counts) */
int malcs_graphs_runs(const malcs_graphs *graphs, uint64_t i, int64_t *runs4);
int malcs_graphs_nodes(const malcs_graphs *graphs, uint64_t i, int64_t *nodes3);
int malcs_graphs_edges(const malcs_graphs *graphs, uint64_t i, int64_t *edges6);
/* Canonical allele of pair i -- stands in for canonical(graph) (algebra/extractor/extractor.py:10-89): BFS over
 * the graph, equally short alternatives merged into one delins up to their lowest common ancestor.
 * count = number of variants; each is {start, end, obs_from, obs_to}: Variant(start, end, observed[obs_from:obs_to]). */
int malcs_graphs_canonical_count(malcs_graphs *graphs, uint64_t i, int64_t *count);
int malcs_graphs_canonical(malcs_graphs *graphs, uint64_t i, int64_t *variants4);
/* Atom masks of `count` graphs of the handle (pairs index[k]) -- stands in for LCSgraph.uniq_atomics()
 * (algebra/lcs/lcs_graph.py:207-217) over DNA: for every position p of the supremal's span [start, end] one byte, bit 0
 * = the atom (p, "") and bit 1 + c = the atom (p, symbol with code c), c = 0 ... 3; two graphs are disjoint
 * (LCSgraph.is_disjoint, :226-228) iff the masks share no bit at any common position.  obs / obs_off: the observed
 * sequences of those pairs as codes 0 ... 3, back to back (count + 1 offsets); graph k's end - start + 1 bytes go to
 * out[out_range[2 k] .. out_range[2 k + 1]). */
int malcs_graphs_atoms(malcs_graphs *graphs, uint64_t count, const uint64_t *index, const uint8_t *obs, const uint64_t *obs_off,
                       const uint64_t *out_range, uint8_t *out);
/* ---- from_variant for many alleles of one reference ------------------------------------------------------
 * Stands in for LCSgraph.from_variant(reference, variants, offset) (algebra/lcs/lcs_graph.py:77-123) called for n
 * alleles of ONE reference: allele i replaces ref[start[i] .. end[i]) by ins[ins_off[i] .. ins_off[i + 1]) (the
 * caller has folded an allele of several variants into that one delins, :95-100).  The widening rounds (:113-123) run
 * inside the library, every round one device batch over the alleles still widening; an allele that changes nothing is
 * the graph of two empty sequences (:101-103).  ref / ins: symbol codes below n_symbols.  The handle keeps one
 * malcs_graphs per round; per allele it gives
 *   where4     {round, index in that round's handle, lo, hi}: the final window is ref[lo .. hi), shift = lo
 *   distance, supremal4 {start, end, obs_from, obs_to}      as the graph handle does
 *   obs_off (n + 1) / observed                               the windows with the alleles in them (codes), back to back:
 *                                                            allele i's supremal sequence is observed[obs_off[i] + obs_from .. + obs_to)
 * and, for the relation matrix (algebra/relations/graph_based.py:51-54, 76-79), in one pass:
 *   key        a number per distinct supremal (start, end, sequence)
 *   symmask    bit (code mod 64) for every symbol of the supremal sequence; *folded != 0 when a code was 64 or more
 *   sup        the supremal sequences (codes) at sup_off[i] (n + 1 offsets, the caller's: cumulative obs_to - obs_from)
 *   atoms      malcs_graphs_atoms of every graph at atom_off[i] (cumulative end - start + 1); DNA codes 0 ... 3 only,
 *              NULL to leave them out */
typedef struct malcs_vgraphs malcs_vgraphs;
int malcs_from_variant_batch(malcs_ctx *ctx, const uint8_t *ref, uint64_t ref_len, uint64_t n, const int64_t *start,
                             const int64_t *end, const uint8_t *ins, const uint64_t *ins_off, int64_t offset, int n_symbols,
                             malcs_vgraphs **graphs);
int malcs_vgraphs_destroy(malcs_vgraphs *graphs);
uint64_t malcs_vgraphs_size(const malcs_vgraphs *graphs);
int malcs_vgraphs_columns(const malcs_vgraphs *graphs, int64_t *where4, int64_t *distance, int64_t *supremal4, uint64_t *obs_off);
int malcs_vgraphs_observed(const malcs_vgraphs *graphs, uint8_t *observed);
int malcs_vgraphs_rounds(const malcs_vgraphs *graphs, int *n_rounds);
/* the handle of one round: borrowed -- it lives as long as `graphs` does and is not the caller's to destroy */
int malcs_vgraphs_round(const malcs_vgraphs *graphs, int round, malcs_graphs **round_graphs, uint64_t *n_pairs);
int malcs_vgraphs_matrix_columns(malcs_vgraphs *graphs, uint32_t *key, uint64_t *symmask, int *folded, const uint64_t *sup_off,
                                 uint8_t *sup, const uint64_t *atom_off, uint8_t *atoms);

/* timing split of the last malcs_lcs_graph_batch on this handle, milliseconds:
 * out[0] forward DP kernel, out[1] reach/extract kernel (HIP events), out[2] D2H + host graph build */
int malcs_graphs_timing(const malcs_graphs *graphs, float *out3);
/* how many of the handle's graphs were constructed on the device (kernel family 2c, malcs_gbuild.hip: ordering of
 * the runs, dominance edges, splits, numbering -- the host only copies the arrays home), and the kernels'
 * milliseconds (HIP events); the others went through the host half.  Either pointer may be NULL. */
int malcs_graphs_device_built(const malcs_graphs *graphs, uint64_t *count, float *ms);
/* For reports: bytes of tie-bit rows (2 bits per stored DP cell) that the bit-vector path's forward pass wrote and its
 * backward pass read back for this handle, and how many pairs took that path (0 / 0 when all pairs took the sweeps). */
int malcs_graphs_tie_bytes(const malcs_graphs *graphs, uint64_t *bytes, uint64_t *pairs);

/* ---- kernel family 3: tiles of the all-vs-all relation matrix -----------------------------------
 * Stands in for graph_based.compare (algebra/relations/graph_based.py:33-79) over all pairs of a list of n LCS graphs
 * (the loop of scripts/benchmark.py:38-52), minus its one DP, which stays with kernel family 1:
 *   create    evaluates the O(1) tests for this rank's tiles on the device: key[i] == key[j] -> EQUIVALENT (:51; the
 *             caller gives equal supremal variants equal keys), Variant.is_disjoint (algebra/variants/variant.py:126-133)
 *             on [start, end) and the symbol bit masks -> DISJOINT (:54); everything else is undecided.
 *             Tiles are tile x tile cells of the upper triangle, numbered row-major and dealt round-robin:
 *             tile number t * world + rank is this rank's tile t.
 *   pending   the undecided pairs (i < j), 2 int64 each.  The caller computes edit_distance over their union windows
 *             (:57-65, malcs_edit_distance_batch), decides :67-79 and hands one relation code per pair to
 *   resolve   (codes follow the reference's Relation order: 0 equivalent, 1 contains, 2 is_contained, 3 overlap,
 *             4 disjoint).
 *   tiles     this rank's label tiles, ceil(n_tiles / world) x tile x tile bytes: the block each rank contributes to
 *             the all-gather (RCCL; the only inter-GPU traffic of the path).
 *   assemble  the full n x n matrix from the gathered blocks (world x ceil(n_tiles / world) x tile x tile, rank-major;
 *             NULL when world == 1): label[j][i] is label[i][j] read from the other side.
 * A symbol mask has one bit per distinct symbol of the variant's inserted sequence; with more than 64 symbols the
 * caller may fold them (then DISJOINT is under-reported and such pairs merely arrive in `pending`). */
typedef struct malcs_relmatrix malcs_relmatrix;
int malcs_relmatrix_create(malcs_ctx *ctx, uint64_t n, const int64_t *start, const int64_t *end, const uint32_t *key,
                           const uint64_t *symmask, uint32_t tile, uint32_t rank, uint32_t world, malcs_relmatrix **out);
int malcs_relmatrix_destroy(malcs_relmatrix *m);
int malcs_relmatrix_pending_count(const malcs_relmatrix *m, uint64_t *count);
int malcs_relmatrix_pending(const malcs_relmatrix *m, int64_t *pairs2);
int malcs_relmatrix_resolve(malcs_relmatrix *m, const uint8_t *codes);
/* pending + the caller's distance batch + resolve in one call, on the device: for every undecided pair the union
 * windows (graph_based.py:57-64) are put together from the reference and the supremals' sequences (codes below
 * n_symbols), their distances computed (:65, kernel family 1), the distance arithmetic applied (:67-74) and what it
 * leaves open decided from the graphs' atom masks (:76-79; malcs_graphs_atoms).  Arrays per variant as in create;
 * sup_off / atom_off have n + 1 entries; variant i's atoms cover positions start[i] ... end[i].  *n_dp = pairs that
 * reached the DP on this rank. */
int malcs_relmatrix_decide(malcs_relmatrix *m, const uint8_t *ref, uint64_t ref_len, const uint8_t *sup, const uint64_t *sup_off,
                           const int64_t *distance, const uint8_t *atoms, const uint64_t *atom_off, int n_symbols, uint64_t *n_dp);
int malcs_relmatrix_tiles(const malcs_relmatrix *m, uint8_t *labels);
/* `full` (n x n labels, row-major) may be any host memory; a copy into pinned memory runs at the link's speed
 * (25 MB for 5 000 variants: 0.5 ms instead of 2.5), and malcs_host_alloc hands out such memory from the library's
 * recycled pinned blocks (malcs_host_free gives a block back to them; neither has a counterpart in the reference,
 * whose results are Python objects). */
int malcs_host_alloc(uint64_t bytes, void **p);
int malcs_host_free(void *p);
int malcs_relmatrix_assemble(malcs_relmatrix *m, const uint8_t *gathered, uint8_t *full);
/* The same two steps for a caller whose collective runs on device memory (RCCL): `tiles_device` hands out the
 * device address of this rank's label block (owned by the handle, valid until destroy; the library's stream has
 * been drained, so any stream may read it), `assemble_device` takes the gathered blocks as a device address on
 * the handle's device (the caller has completed the collective that wrote them).  No host staging of the tiles. */
int malcs_relmatrix_tiles_device(const malcs_relmatrix *m, const void **d_labels, uint64_t *bytes);
int malcs_relmatrix_assemble_device(malcs_relmatrix *m, const void *d_gathered, uint8_t *full);

/* ---- the one collective: all-gather of the label tiles over RCCL ---------------------------------------
 * NO COUNTERPART IN THE REFERENCE (it is single-threaded, single-process).  One process per GPU; the ranks of a job
 * meet through a 128-byte id: rank 0 calls malcs_comm_unique_id and carries the bytes to the others (any channel:
 * algebra_amd/comm.py uses a TCP socket at MASTER_ADDR), then every rank calls malcs_comm_create on its own context.
 * librccl.so is loaded when the first of these is called (MALCS_RCCL_LIB overrides the name); the collective runs
 * on the context's stream on the library's own buffers.
 *   all_gather       `bytes` bytes at device address d_send of every rank -> world * bytes at d_recv (rank-major);
 *                    returns when it is complete
 *   malcs_relmatrix_assemble_comm   tiles (above) -> ncclAllGather of every rank's label block -> assemble: every
 *                    rank ends with the full n x n matrix in `full`; nothing is staged through the host. */
typedef struct malcs_comm malcs_comm;
int malcs_comm_unique_id(void *id128);
int malcs_comm_create(malcs_ctx *ctx, const void *id128, int rank, int world, malcs_comm **comm);
int malcs_comm_destroy(malcs_comm *comm);
int malcs_comm_info(const malcs_comm *comm, int *rank, int *world);
int malcs_comm_all_gather(malcs_comm *comm, const void *d_send, void *d_recv, uint64_t bytes);
int malcs_relmatrix_assemble_comm(malcs_relmatrix *m, malcs_comm *comm, uint8_t *full);

#ifdef __cplusplus
}
#endif
#endif /* MALCS_H */
