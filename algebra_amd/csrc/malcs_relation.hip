// malcs_relation.hip -- kernel family 3: tiles of the all-vs-all relation matrix.
//
// Stands in for the parts of graph_based.compare (reference algebra/relations/graph_based.py:33-79) that need no
// DP, evaluated for whole tiles of the matrix over a list of LCS graphs (the shape of scripts/benchmark.py:38-52):
//   :51     supremal == supremal                      -> EQUIVALENT   (the caller numbers equal supremals alike: `key`)
//   :54     supremal.is_disjoint(supremal)            -> DISJOINT     (algebra/variants/variant.py:126-133: intervals
//           that share no position and are either apart or insert/delete no common symbol; `symmask` holds one bit
//           per symbol of the inserted sequence)
//   else    undecided (255): the pair goes to the caller's batch of union-window distances (:57-65, kernel family 1),
//           whose verdicts (:67-79) come back through malcs_relmatrix_resolve.
// and for the bookkeeping around them: upper-triangular tiles dealt round-robin to ranks, the label tiles of a rank
// as one contiguous block for the all-gather, and the assembly of the full matrix (mirror of a label = the relation
// read from the other side: CONTAINS <-> IS_CONTAINED).
//
// One thread per cell, consecutive threads along a tile row: every access to the label tiles is coalesced and the
// per-variant arrays (a few hundred KB) stay in L2.  HBM-bound and tiny: n^2 / 2 bytes written once, n^2 bytes for
// the full matrix.
#include "malcs_internal.hpp"
#include <algorithm>
#include <new>

namespace {

constexpr uint8_t EQUIVALENT = 0, CONTAINS = 1, IS_CONTAINED = 2, DISJOINT = 4, UNDECIDED = 255;

// tile number idx (row-major over the upper triangle of a blocks x blocks grid) -> (block row, block col)
__host__ __device__ inline void tile_coords(uint32_t idx, uint32_t blocks, uint32_t &bi, uint32_t &bj)
{
    uint32_t row = 0, left = idx;
    while (left >= blocks - row) { left -= blocks - row; row++; }
    bi = row; bj = row + left;
}
__host__ __device__ inline uint32_t tile_index(uint32_t bi, uint32_t bj, uint32_t blocks)
{
    return bi * blocks - bi * (bi - 1u) / 2u + (bj - bi);
}

// grid (tiles of this rank, cells of a tile / 256)
__global__ void __launch_bounds__(256)
k_relation_tiles(uint32_t n, const int64_t *__restrict__ start, const int64_t *__restrict__ end,
                 const uint32_t *__restrict__ key, const uint64_t *__restrict__ symmask, uint32_t tile, uint32_t blocks,
                 uint32_t rank, uint32_t world, uint8_t *__restrict__ labels, unsigned long long *__restrict__ n_pending)
{
    const uint32_t t = blockIdx.x, cell = blockIdx.y * 256u + threadIdx.x;
    if (cell >= tile * tile) return;
    uint32_t bi, bj;
    tile_coords(t * world + rank, blocks, bi, bj);
    const uint32_t r = cell / tile, c = cell % tile, i = bi * tile + r, j = bj * tile + c;
    uint8_t label = EQUIVALENT;                                  // cells outside the matrix and the mirrored half of
    if (i < n && j < n && !(bi == bj && r > c)) {                 // a diagonal tile stay 0
        const int64_t si = start[i], ei = end[i], sj = start[j], ej = end[j];
        if (key[i] == key[j]) label = EQUIVALENT;                                            // :51
        else if (!(sj < ei && si < ej) && (sj > ei || si > ej || (symmask[i] & symmask[j]) == 0)) label = DISJOINT;   // :54
        else label = UNDECIDED;
    }
    labels[(size_t)t * tile * tile + cell] = label;
    // one atomic per wave
    const uint64_t votes = __ballot(label == UNDECIDED);
    if (votes && (threadIdx.x & 63u) == (uint32_t)__builtin_ctzll(votes)) atomicAdd(n_pending, (unsigned long long)__builtin_popcountll(votes));
}

__global__ void __launch_bounds__(256)
k_relation_collect(const uint8_t *__restrict__ labels, uint64_t n_cells, unsigned long long *__restrict__ cursor,
                   uint64_t cap, uint64_t *__restrict__ cells)
{
    const uint64_t cell = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (cell >= n_cells || labels[cell] != UNDECIDED) return;
    const unsigned long long slot = atomicAdd(cursor, 1ull);
    if (slot < cap) cells[slot] = cell;
}

__global__ void __launch_bounds__(256)
k_relation_resolve(uint8_t *__restrict__ labels, const uint64_t *__restrict__ cells, const uint8_t *__restrict__ codes, uint64_t count)
{
    const uint64_t k = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (k < count) labels[cells[k]] = codes[k];
}

// full[i][j]: grid (n, ceil(n / 256)); the tile of (min, max) holds the label, read from the other side when i > j
__global__ void __launch_bounds__(256)
k_relation_assemble(uint32_t n, uint32_t tile, uint32_t blocks, uint32_t world, uint32_t per_rank,
                    const uint8_t *__restrict__ gathered, uint8_t *__restrict__ full)
{
    const uint32_t i = blockIdx.x, j = blockIdx.y * 256u + threadIdx.x;
    if (j >= n) return;
    uint8_t v = EQUIVALENT;
    if (i != j) {
        const uint32_t lo = min(i, j), hi = max(i, j);
        const uint32_t idx = tile_index(lo / tile, hi / tile, blocks);
        const size_t slot = (size_t)(idx % world) * per_rank + idx / world;
        v = gathered[slot * tile * tile + (size_t)(lo % tile) * tile + hi % tile];
        if (i > j) v = v == CONTAINS ? IS_CONTAINED : (v == IS_CONTAINED ? CONTAINS : v);
    }
    full[(size_t)i * n + j] = v;
}

// ---- the undecided pairs, decided on the device ----------------------------------------------------------
// pending cell k -> its variants (i, j) and the lengths of its two union-window sequences (graph_based.py:57-64):
// window [min(start), max(end)), each side = reference up to its supremal + the supremal's sequence + reference after it
__global__ void __launch_bounds__(256)
k_relation_window_lens(const uint64_t *__restrict__ cells, uint64_t count, uint32_t tile, uint32_t blocks, uint32_t rank, uint32_t world,
                       const int64_t *__restrict__ start, const int64_t *__restrict__ end, const uint64_t *__restrict__ sup_off,
                       uint32_t *__restrict__ ij, uint32_t *__restrict__ lens)
{
    const uint64_t k = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (k >= count) return;
    const uint64_t tile_cells = (uint64_t)tile * tile, cell = cells[k], t = cell / tile_cells, in = cell % tile_cells;
    uint32_t bi, bj;
    tile_coords((uint32_t)(t * world + rank), blocks, bi, bj);
    const uint32_t i = bi * tile + (uint32_t)(in / tile), j = bj * tile + (uint32_t)(in % tile);
    ij[2 * k] = i; ij[2 * k + 1] = j;
    const int64_t ws = min(start[i], start[j]), we = max(end[i], end[j]);
    lens[2 * k] = (uint32_t)((start[i] - ws) + (int64_t)(sup_off[i + 1] - sup_off[i]) + (we - end[i]));
    lens[2 * k + 1] = (uint32_t)((start[j] - ws) + (int64_t)(sup_off[j + 1] - sup_off[j]) + (we - end[j]));
}

// one workgroup of 64 per window sequence (2 k: the side of variant i, 2 k + 1: of variant j)
__global__ void __launch_bounds__(64)
k_relation_window_fill(const uint32_t *__restrict__ ij, const int64_t *__restrict__ start, const int64_t *__restrict__ end,
                       const uint8_t *__restrict__ ref, const uint8_t *__restrict__ sup, const uint64_t *__restrict__ sup_off,
                       const uint64_t *__restrict__ out_off, uint8_t *__restrict__ out)
{
    const uint64_t q = blockIdx.x, k = q >> 1;
    const uint32_t i = ij[2 * k], j = ij[2 * k + 1], v = (q & 1) ? j : i;
    const int64_t ws = min(start[i], start[j]), we = max(end[i], end[j]);
    const int64_t pre = start[v] - ws, mid = (int64_t)(sup_off[v + 1] - sup_off[v]), post = we - end[v];
    uint8_t *o = out + out_off[q];
    for (int64_t p = threadIdx.x; p < pre + mid + post; p += 64)
        o[p] = p < pre ? ref[ws + p] : (p < pre + mid ? sup[sup_off[v] + (uint64_t)(p - pre)] : ref[end[v] + (p - pre - mid)]);
}

// :67-79 for pending pair k: the three distances, then the atom masks; the verdict goes straight into the label tile
__global__ void __launch_bounds__(256)
k_relation_verdict(const uint64_t *__restrict__ cells, const uint32_t *__restrict__ ij, uint64_t count, const int32_t *__restrict__ dist,
                   const int64_t *__restrict__ distance, const int64_t *__restrict__ start, const uint8_t *__restrict__ atoms,
                   const uint64_t *__restrict__ atom_off, uint8_t *__restrict__ labels)
{
    const uint64_t k = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (k >= count) return;
    const uint32_t i = ij[2 * k], j = ij[2 * k + 1];
    const int64_t li = distance[i], rj = distance[j], d = dist[k];
    uint8_t code;
    if (li + rj == d) code = DISJOINT;
    else if (li - rj == d) code = CONTAINS;
    else if (rj - li == d) code = IS_CONTAINED;
    else {
        // LCSgraph.is_disjoint: no atom in common at any position both graphs have atoms for
        const int64_t lo_i = start[i], lo_j = start[j];
        const int64_t n_i = (int64_t)(atom_off[i + 1] - atom_off[i]), n_j = (int64_t)(atom_off[j + 1] - atom_off[j]);
        const int64_t lo = max(lo_i, lo_j), hi = min(lo_i + n_i, lo_j + n_j);
        const uint8_t *a = atoms + atom_off[i] + (lo - lo_i), *b = atoms + atom_off[j] + (lo - lo_j);
        uint8_t any = 0;
        for (int64_t p = 0; p < hi - lo; p++) any |= a[p] & b[p];
        code = any ? (uint8_t)3 : DISJOINT;                          // 3 = OVERLAP
    }
    labels[cells[k]] = code;
}

template <typename T> struct dev {
    T *p = nullptr;
    ~dev() { if (p) hipFree(p); }
    int alloc(size_t count) { return hipMalloc(&p, (count ? count : 1) * sizeof(T)) == hipSuccess ? 0 : -1; }
};

}  // namespace

struct malcs_relmatrix {
    malcs_ctx *ctx = nullptr;
    uint32_t n = 0, tile = 0, blocks = 0, rank = 0, world = 1, n_tiles = 0, mine = 0, per_rank = 0;
    dev<uint8_t> labels;               // this rank's tiles, per_rank x tile x tile (unused slots zero)
    dev<uint64_t> cells;               // undecided cells (index into labels)
    dev<int64_t> start, end;           // the variants' intervals (kept for malcs_relmatrix_decide)
    uint64_t n_pending = 0;
    std::vector<uint64_t> host_cells;
};

extern "C" {

int malcs_relmatrix_create(malcs_ctx *ctx, uint64_t n, const int64_t *start, const int64_t *end, const uint32_t *key,
                           const uint64_t *symmask, uint32_t tile, uint32_t rank, uint32_t world, malcs_relmatrix **out)
{
    if (!out) { malcs_set_error("out is NULL"); return MALCS_ERR_ARG; }
    *out = nullptr;
    if (!ctx || (n && (!start || !end || !key || !symmask)) || tile == 0 || tile > 2048 || world == 0 || rank >= world ||
        n > 0x7fffffffull) {
        malcs_set_error("bad argument to malcs_relmatrix_create");
        return MALCS_ERR_ARG;
    }
    MALCS_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    malcs_relmatrix *h = new (std::nothrow) malcs_relmatrix();
    if (!h) { malcs_set_error("out of host memory"); return MALCS_ERR_NOMEM; }
    h->ctx = ctx; h->n = (uint32_t)n; h->tile = tile; h->rank = rank; h->world = world;
    h->blocks = (uint32_t)((n + tile - 1) / tile);
    const uint64_t n_tiles = (uint64_t)h->blocks * (h->blocks + 1) / 2;
    if (n_tiles > 0x7fffffffull) { delete h; malcs_set_error("too many tiles"); return MALCS_ERR_TOO_LONG; }
    h->n_tiles = (uint32_t)n_tiles;
    h->mine = h->n_tiles > rank ? (h->n_tiles - rank + world - 1) / world : 0;
    h->per_rank = (h->n_tiles + world - 1) / world;
    const size_t tile_cells = (size_t)tile * tile, n_cells = (size_t)h->per_rank * tile_cells;
    dev<int64_t> &d_start = h->start, &d_end = h->end;
    dev<uint32_t> d_key;
    dev<uint64_t> d_mask;
    dev<unsigned long long> d_count;
    int rc = MALCS_OK;
    do {
        if (h->labels.alloc(n_cells) || d_start.alloc(n) || d_end.alloc(n) || d_key.alloc(n) || d_mask.alloc(n) || d_count.alloc(2)) {
            malcs_set_error("device allocation failed (%llu label bytes)", (unsigned long long)n_cells);
            rc = MALCS_ERR_NOMEM; break;
        }
#define RM_HIP(call) { hipError_t e_ = (call); if (e_ != hipSuccess) { malcs_set_error("%s failed: %s", #call, hipGetErrorString(e_)); rc = MALCS_ERR_HIP; break; } }
        RM_HIP(hipMemsetAsync(h->labels.p, 0, n_cells ? n_cells : 1, st));
        RM_HIP(hipMemsetAsync(d_count.p, 0, 16, st));
        if (n) {
            RM_HIP(malcs_h2d(st, d_start.p, start, n * 8));
            RM_HIP(malcs_h2d(st, d_end.p, end, n * 8));
            RM_HIP(malcs_h2d(st, d_key.p, key, n * 4));
            RM_HIP(malcs_h2d(st, d_mask.p, symmask, n * 8));
        }
        if (h->mine) {
            hipLaunchKernelGGL(k_relation_tiles, dim3(h->mine, (uint32_t)((tile_cells + 255) / 256)), dim3(256), 0, st, h->n, d_start.p,
                               d_end.p, d_key.p, d_mask.p, tile, h->blocks, rank, world, h->labels.p, d_count.p);
            RM_HIP(hipGetLastError());
        }
        unsigned long long count = 0;
        RM_HIP(malcs_d2h(st, &count, d_count.p, 8));
        RM_HIP(malcs_sync(st));
        h->n_pending = count;
        if (count) {
            if (h->cells.alloc(count)) { malcs_set_error("device allocation failed (%llu undecided pairs)", count); rc = MALCS_ERR_NOMEM; break; }
            const uint64_t scan = (uint64_t)h->mine * tile_cells;
            hipLaunchKernelGGL(k_relation_collect, dim3((uint32_t)((scan + 255) / 256)), dim3(256), 0, st, h->labels.p, scan, d_count.p + 1,
                               (uint64_t)count, h->cells.p);
            RM_HIP(hipGetLastError());
            h->host_cells.resize(count);
            RM_HIP(malcs_d2h(st, h->host_cells.data(), h->cells.p, count * 8));
            RM_HIP(malcs_sync(st));
        }
    } while (0);
    if (rc) { delete h; return rc; }
    *out = h;
    return MALCS_OK;
}

int malcs_relmatrix_destroy(malcs_relmatrix *h) { delete h; return MALCS_OK; }

int malcs_relmatrix_pending_count(const malcs_relmatrix *h, uint64_t *count)
{
    if (!h || !count) { malcs_set_error("NULL argument"); return MALCS_ERR_ARG; }
    *count = h->n_pending;
    return MALCS_OK;
}

int malcs_relmatrix_pending(const malcs_relmatrix *h, int64_t *pairs2)
{
    if (!h || (h->n_pending && !pairs2)) { malcs_set_error("NULL argument"); return MALCS_ERR_ARG; }
    const uint64_t tile_cells = (uint64_t)h->tile * h->tile;
    for (uint64_t k = 0; k < h->n_pending; k++) {
        const uint64_t cell = h->host_cells[k], t = cell / tile_cells, in = cell % tile_cells;
        uint32_t bi, bj;
        tile_coords((uint32_t)(t * h->world + h->rank), h->blocks, bi, bj);
        pairs2[2 * k] = (int64_t)bi * h->tile + (int64_t)(in / h->tile);
        pairs2[2 * k + 1] = (int64_t)bj * h->tile + (int64_t)(in % h->tile);
    }
    return MALCS_OK;
}

int malcs_relmatrix_resolve(malcs_relmatrix *h, const uint8_t *codes)
{
    if (!h || (h->n_pending && !codes)) { malcs_set_error("NULL argument"); return MALCS_ERR_ARG; }
    if (!h->n_pending) return MALCS_OK;
    for (uint64_t k = 0; k < h->n_pending; k++)
        if (codes[k] > DISJOINT) { malcs_set_error("code %u of pair %llu is not a relation", codes[k], (unsigned long long)k); return MALCS_ERR_ARG; }
    MALCS_HIP(hipSetDevice(h->ctx->device));
    hipStream_t st = h->ctx->stream;
    dev<uint8_t> d_codes;
    if (d_codes.alloc(h->n_pending)) { malcs_set_error("device allocation failed"); return MALCS_ERR_NOMEM; }
    MALCS_HIP(malcs_h2d(st, d_codes.p, codes, h->n_pending));
    hipLaunchKernelGGL(k_relation_resolve, dim3((uint32_t)((h->n_pending + 255) / 256)), dim3(256), 0, st, h->labels.p, h->cells.p,
                       d_codes.p, h->n_pending);
    MALCS_HIP(hipGetLastError());
    MALCS_HIP(malcs_sync(st));
    return MALCS_OK;
}

int malcs_relmatrix_decide(malcs_relmatrix *h, const uint8_t *ref, uint64_t ref_len, const uint8_t *sup, const uint64_t *sup_off,
                           const int64_t *distance, const uint8_t *atoms, const uint64_t *atom_off, int n_symbols, uint64_t *n_dp)
{
    if (!h || !sup_off || !distance || !atom_off || (ref_len && !ref)) { malcs_set_error("NULL argument"); return MALCS_ERR_ARG; }
    if (n_dp) *n_dp = h->n_pending;
    if (!h->n_pending) return MALCS_OK;
    malcs_ctx *ctx = h->ctx;
    MALCS_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const uint64_t count = h->n_pending, n = h->n;
    if (sup_off[n] && !sup) { malcs_set_error("NULL argument"); return MALCS_ERR_ARG; }
    if (atom_off[n] && !atoms) { malcs_set_error("NULL argument"); return MALCS_ERR_ARG; }
    pool_scoped<uint8_t> d_ref(ctx), d_sup(ctx), d_atoms(ctx), d_syms(ctx);
    pool_scoped<uint64_t> d_sup_off(ctx), d_atom_off(ctx), d_out_off(ctx);
    pool_scoped<int64_t> d_distance(ctx);
    pool_scoped<uint32_t> d_ij(ctx), d_lens(ctx);
    MALCS_HIP(d_ref.alloc(ref_len + 1)); MALCS_HIP(d_sup.alloc(sup_off[n] + 1)); MALCS_HIP(d_atoms.alloc(atom_off[n] + 1));
    MALCS_HIP(d_sup_off.alloc(n + 1)); MALCS_HIP(d_atom_off.alloc(n + 1)); MALCS_HIP(d_distance.alloc(n));
    MALCS_HIP(d_ij.alloc(2 * count)); MALCS_HIP(d_lens.alloc(2 * count)); MALCS_HIP(d_out_off.alloc(2 * count));
    if (ref_len) MALCS_HIP(malcs_h2d(st, d_ref.p, ref, ref_len));
    if (sup_off[n]) MALCS_HIP(malcs_h2d(st, d_sup.p, sup, sup_off[n]));
    if (atom_off[n]) MALCS_HIP(malcs_h2d(st, d_atoms.p, atoms, atom_off[n]));
    MALCS_HIP(malcs_h2d(st, d_sup_off.p, sup_off, (n + 1) * 8));
    MALCS_HIP(malcs_h2d(st, d_atom_off.p, atom_off, (n + 1) * 8));
    MALCS_HIP(malcs_h2d(st, d_distance.p, distance, n * 8));
    const dim3 grid((uint32_t)((count + 255) / 256)), b256(256);
    hipLaunchKernelGGL(k_relation_window_lens, grid, b256, 0, st, h->cells.p, count, h->tile, h->blocks, h->rank, h->world, h->start.p, h->end.p,
                       d_sup_off.p, d_ij.p, d_lens.p);
    MALCS_HIP(hipGetLastError());
    std::vector<uint32_t> lens(2 * count);
    MALCS_HIP(malcs_d2h(st, lens.data(), d_lens.p, 2 * count * sizeof(uint32_t)));
    MALCS_HIP(malcs_sync(st));
    // the windows back to back: pair k's two sequences at off[2 k], off[2 k + 1]
    std::vector<uint64_t> off(2 * count);
    std::vector<uint64_t> a_off(count), b_off(count);
    std::vector<uint32_t> a_len(count), b_len(count);
    uint64_t total = 0;
    for (uint64_t q = 0; q < 2 * count; q++) { off[q] = total; total += lens[q]; }
    for (uint64_t k = 0; k < count; k++) { a_off[k] = off[2 * k]; a_len[k] = lens[2 * k]; b_off[k] = off[2 * k + 1]; b_len[k] = lens[2 * k + 1]; }
    MALCS_HIP(d_syms.alloc(total + 1));
    MALCS_HIP(malcs_h2d(st, d_out_off.p, off.data(), 2 * count * 8));
    hipLaunchKernelGGL(k_relation_window_fill, dim3((uint32_t)(2 * count)), dim3(64), 0, st, d_ij.p, h->start.p, h->end.p, d_ref.p, d_sup.p,
                       d_sup_off.p, d_out_off.p, d_syms.p);
    MALCS_HIP(hipGetLastError());
    if (2 * count * 8 > (1u << 20)) MALCS_HIP(malcs_sync(st));
    malcs_batch *b = nullptr;
    int rc = malcs_batch_create_device(ctx, d_syms.p, total, a_off.data(), a_len.data(), b_off.data(), b_len.data(), count, n_symbols, &b);
    if (rc) return rc;
    rc = malcs_dist_launch(b);
    if (!rc) {
        hipLaunchKernelGGL(k_relation_verdict, grid, b256, 0, st, h->cells.p, d_ij.p, count, b->d_dist, d_distance.p, h->start.p, d_atoms.p,
                           d_atom_off.p, h->labels.p);
        if (hipGetLastError() != hipSuccess) { malcs_set_error("verdict kernel launch failed"); rc = MALCS_ERR_HIP; }
    }
    malcs_batch_destroy(b);                    // waits for the stream: the verdicts are in the tiles
    return rc;
}

int malcs_relmatrix_tiles(const malcs_relmatrix *h, uint8_t *labels)
{
    if (!h || !labels) { malcs_set_error("NULL argument"); return MALCS_ERR_ARG; }
    MALCS_HIP(hipSetDevice(h->ctx->device));
    const size_t bytes = (size_t)h->per_rank * h->tile * h->tile;
    if (bytes) MALCS_HIP(malcs_d2h(h->ctx->stream, labels, h->labels.p, bytes));
    MALCS_HIP(malcs_sync(h->ctx->stream));
    return MALCS_OK;
}

int malcs_relmatrix_tiles_device(const malcs_relmatrix *h, const void **d_labels, uint64_t *bytes)
{
    if (!h || !d_labels || !bytes) { malcs_set_error("NULL argument"); return MALCS_ERR_ARG; }
    MALCS_HIP(hipSetDevice(h->ctx->device));
    MALCS_HIP(malcs_sync(h->ctx->stream));       // resolve() may still be scattering verdicts
    *d_labels = h->labels.p;
    *bytes = (uint64_t)h->per_rank * h->tile * h->tile;
    return MALCS_OK;
}

// `d_src`: the gathered blocks on the device, or NULL for this rank's own block (world == 1)
static int relmatrix_assemble(malcs_relmatrix *h, const uint8_t *d_src, uint8_t *full)
{
    hipStream_t st = h->ctx->stream;
    pool_scoped<uint8_t> d_full(h->ctx);           // from the context's pool: no hipMalloc / hipFree of n x n bytes per matrix
    if (d_full.alloc((size_t)h->n * h->n) != hipSuccess) { (void)hipGetLastError(); malcs_set_error("device allocation failed (%u x %u labels)", h->n, h->n); return MALCS_ERR_NOMEM; }
    hipLaunchKernelGGL(k_relation_assemble, dim3(h->n, (h->n + 255u) / 256u), dim3(256), 0, st, h->n, h->tile, h->blocks, h->world,
                       h->per_rank, d_src ? d_src : h->labels.p, d_full.p);
    MALCS_HIP(hipGetLastError());
    MALCS_HIP(malcs_d2h(st, full, d_full.p, (size_t)h->n * h->n));
    MALCS_HIP(malcs_sync(st));
    return MALCS_OK;
}

int malcs_relmatrix_assemble(malcs_relmatrix *h, const uint8_t *gathered, uint8_t *full)
{
    if (!h || (h->n && !full) || (h->world > 1 && !gathered)) { malcs_set_error("NULL argument"); return MALCS_ERR_ARG; }
    if (!h->n) return MALCS_OK;
    MALCS_HIP(hipSetDevice(h->ctx->device));
    const size_t bytes = (size_t)h->world * h->per_rank * h->tile * h->tile;
    dev<uint8_t> d_gathered;
    if (gathered) {
        if (d_gathered.alloc(bytes)) { malcs_set_error("device allocation failed (%llu gathered label bytes)", (unsigned long long)bytes); return MALCS_ERR_NOMEM; }
        MALCS_HIP(malcs_h2d(h->ctx->stream, d_gathered.p, gathered, bytes));
    }
    return relmatrix_assemble(h, d_gathered.p, full);
}

int malcs_relmatrix_assemble_comm(malcs_relmatrix *h, malcs_comm *comm, uint8_t *full)
{
    if (!h || !comm || (h->n && !full)) { malcs_set_error("NULL argument"); return MALCS_ERR_ARG; }
    if (malcs_comm_ctx(comm) != h->ctx || (uint32_t)malcs_comm_world(comm) != h->world || (uint32_t)malcs_comm_rank(comm) != h->rank) {
        malcs_set_error("the communicator (rank %d of %d) does not match the tiles (rank %u of %u) or their context",
                        malcs_comm_rank(comm), malcs_comm_world(comm), h->rank, h->world);
        return MALCS_ERR_ARG;
    }
    if (!h->n) return MALCS_OK;
    MALCS_HIP(hipSetDevice(h->ctx->device));
    // stream order does the rest: resolve() scattered the verdicts on this stream, the collective reads this rank's
    // block where they were written and writes everybody's where the assembly kernel reads them
    const size_t block = (size_t)h->per_rank * h->tile * h->tile;
    dev<uint8_t> d_gathered;
    if (d_gathered.alloc(block * h->world)) { malcs_set_error("device allocation failed (%llu gathered label bytes)", (unsigned long long)(block * h->world)); return MALCS_ERR_NOMEM; }
    const int rc = malcs_comm_all_gather_async(comm, h->labels.p, d_gathered.p, block);
    if (rc) return rc;
    return relmatrix_assemble(h, d_gathered.p, full);
}

int malcs_relmatrix_assemble_device(malcs_relmatrix *h, const void *d_gathered, uint8_t *full)
{
    if (!h || (h->n && !full) || !d_gathered) { malcs_set_error("NULL argument"); return MALCS_ERR_ARG; }
    if (!h->n) return MALCS_OK;
    MALCS_HIP(hipSetDevice(h->ctx->device));
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, d_gathered) != hipSuccess || attr.type != hipMemoryTypeDevice || attr.device != h->ctx->device) {
        (void)hipGetLastError();
        malcs_set_error("gathered labels are not device memory of device %d", h->ctx->device);
        return MALCS_ERR_ARG;
    }
    return relmatrix_assemble(h, (const uint8_t *)d_gathered, full);
}

}  // extern "C"
