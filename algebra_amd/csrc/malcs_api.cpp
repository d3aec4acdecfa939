// malcs_api.cpp -- the C-ABI of libmalcs (include/malcs.h): contexts, batches, family-1 entry points.
#include "malcs_internal.hpp"
#include "malcs_hostpack.hpp"
#include <algorithm>
#include <thread>
#include <mutex>
#include <cstdarg>
#include <cstring>
#include <new>

static thread_local char g_last_error[512] = "";

void malcs_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_last_error, sizeof(g_last_error), fmt, ap);
    va_end(ap);
}

extern "C" {

int malcs_version(void) { return MALCS_VERSION; }
const char *malcs_last_error(void) { return g_last_error; }

int malcs_device_count(int *count)
{
    if (!count) { malcs_set_error("count is NULL"); return MALCS_ERR_ARG; }
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { *count = 0; malcs_set_error("hipGetDeviceCount: %s", hipGetErrorString(e)); return MALCS_ERR_HIP; }
    *count = n;
    return MALCS_OK;
}

}  // extern "C"

// ---- staged copies ---------------------------------------------------------------------------------
namespace {
constexpr size_t STAGE_BYTES = 16u << 20, STAGE_MAX_COPY = 1u << 20;
std::mutex g_ctx_mutex;
std::vector<malcs_ctx *> g_ctxs;

malcs_ctx *ctx_of(hipStream_t st)
{
    std::lock_guard<std::mutex> lock(g_ctx_mutex);
    for (malcs_ctx *c : g_ctxs) if (c->stream == st) return c;
    return nullptr;
}

hipError_t drain(malcs_ctx *c)
{
    const hipError_t e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess) for (const staged_copy &s : c->staged) memcpy(s.dst, s.src, s.bytes);
    c->staged.clear();
    c->stage_used = 0;
    return e;
}

char *stage_slot(malcs_ctx *c, size_t bytes, hipError_t &e)
{
    const size_t need = (bytes + 255u) & ~(size_t)255u;
    e = hipSuccess;
    if (c->stage_used + need > c->stage_cap) { e = drain(c); if (e != hipSuccess) return nullptr; }
    char *p = c->stage + c->stage_used;
    c->stage_used += need;
    return p;
}
}  // namespace

hipError_t malcs_h2d(hipStream_t st, void *dst, const void *src, size_t bytes)
{
    if (!bytes) return hipSuccess;
    malcs_ctx *c = bytes <= STAGE_MAX_COPY ? ctx_of(st) : nullptr;
    if (!c || !c->stage) return hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, st);
    hipError_t e;
    char *slot = stage_slot(c, bytes, e);
    if (!slot) return e;
    memcpy(slot, src, bytes);
    return hipMemcpyAsync(dst, slot, bytes, hipMemcpyHostToDevice, st);
}

hipError_t malcs_d2h(hipStream_t st, void *dst, const void *src, size_t bytes)
{
    if (!bytes) return hipSuccess;
    malcs_ctx *c = bytes <= STAGE_MAX_COPY ? ctx_of(st) : nullptr;
    if (!c || !c->stage) return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, st);
    hipError_t e;
    char *slot = stage_slot(c, bytes, e);
    if (!slot) return e;
    e = hipMemcpyAsync(slot, src, bytes, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) c->staged.push_back({dst, slot, bytes});
    return e;
}

hipError_t malcs_sync(hipStream_t st)
{
    malcs_ctx *c = ctx_of(st);
    return c ? drain(c) : hipStreamSynchronize(st);
}

// ---- pinned host memory, process-wide ----------------------------------------------------------------
namespace {
struct pinned_block { void *p; size_t bytes; bool used; };
std::mutex g_pinned_mutex;
std::vector<pinned_block> g_pinned;
}

void *malcs_pinned_alloc(size_t bytes)
{
    bytes = (bytes + 4095u) & ~(size_t)4095u;
    std::lock_guard<std::mutex> lock(g_pinned_mutex);
    size_t best = (size_t)-1;
    for (size_t i = 0; i < g_pinned.size(); i++) {
        const pinned_block &k = g_pinned[i];
        if (!k.used && k.bytes >= bytes && k.bytes <= 2 * bytes + (1u << 20) && (best == (size_t)-1 || k.bytes < g_pinned[best].bytes)) best = i;
    }
    if (best != (size_t)-1) { g_pinned[best].used = true; return g_pinned[best].p; }
    // nothing fits: let go of what lies idle before asking for more (pinned memory is a scarce kind)
    size_t idle = 0;
    for (const pinned_block &k : g_pinned) if (!k.used) idle += k.bytes;
    if (idle > (2ull << 30) || g_pinned.size() > 64) {
        size_t kept = 0;
        for (size_t i = 0; i < g_pinned.size(); i++) { if (g_pinned[i].used) g_pinned[kept++] = g_pinned[i]; else (void)hipHostFree(g_pinned[i].p); }
        g_pinned.resize(kept);
    }
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes, hipHostMallocPortable) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    g_pinned.push_back({p, bytes, true});
    return p;
}

void malcs_pinned_free(void *p)
{
    if (!p) return;
    std::lock_guard<std::mutex> lock(g_pinned_mutex);
    for (pinned_block &k : g_pinned) if (k.p == p) { k.used = false; return; }
}

extern "C" int malcs_host_alloc(uint64_t bytes, void **p)
{
    if (!p) { malcs_set_error("NULL argument"); return MALCS_ERR_ARG; }
    *p = malcs_pinned_alloc(bytes ? bytes : 1);
    if (!*p) { malcs_set_error("no pinned host memory for %llu bytes", (unsigned long long)bytes); return MALCS_ERR_NOMEM; }
    return MALCS_OK;
}

extern "C" int malcs_host_free(void *p)
{
    malcs_pinned_free(p);
    return MALCS_OK;
}

// ---- pooled device memory ---------------------------------------------------------------------------
hipError_t malcs_pool_alloc(malcs_ctx *c, void **p, size_t bytes)
{
    if (!bytes) bytes = 256;
    bytes = (bytes + 255u) & ~(size_t)255u;
    size_t best = (size_t)-1;
    for (size_t i = 0; i < c->pool.size(); i++) {                // smallest free block that is not wastefully large
        const malcs_ctx::pool_block &k = c->pool[i];
        if (!k.used && k.bytes >= bytes && k.bytes <= 2 * bytes + (1u << 20) && (best == (size_t)-1 || k.bytes < c->pool[best].bytes)) best = i;
    }
    if (best != (size_t)-1) { c->pool[best].used = true; *p = c->pool[best].p; return hipSuccess; }
    hipError_t e = hipMalloc(p, bytes);
    if (e != hipSuccess) {                                       // make room from what the pool holds and try once more
        (void)hipGetLastError();
        malcs_pool_trim(c);
        e = hipMalloc(p, bytes);
        if (e != hipSuccess) return e;
    }
    c->pool.push_back({*p, bytes, true});
    c->pool_bytes += bytes;
    return hipSuccess;
}

void malcs_pool_free(malcs_ctx *c, void *p)
{
    if (!p) return;
    for (malcs_ctx::pool_block &k : c->pool) if (k.p == p) { k.used = false; break; }
    // a context that has seen one huge batch does not sit on its memory for ever
    size_t idle = 0;
    for (const malcs_ctx::pool_block &k : c->pool) if (!k.used) idle += k.bytes;
    if (idle > (8ull << 30) || c->pool.size() > 256) malcs_pool_trim(c);
}

void malcs_pool_trim(malcs_ctx *c)
{
    if (c->stream) (void)hipStreamSynchronize(c->stream);        // queued work may still read a released block
    size_t kept = 0;
    for (size_t i = 0; i < c->pool.size(); i++) {
        if (c->pool[i].used) { c->pool[kept++] = c->pool[i]; continue; }
        (void)hipFree(c->pool[i].p);
        c->pool_bytes -= c->pool[i].bytes;
    }
    c->pool.resize(kept);
}

// everything a context owns, released in reverse order of creation (also on the error paths of malcs_ctx_create)
malcs_workers *malcs_ctx_workers(malcs_ctx *ctx)
{
    if (!ctx->workers) {
        int threads = 16;
        if (const char *e = getenv("MALCS_PACK_THREADS")) threads = std::max(1, atoi(e));
        threads = std::min<int>(threads, (int)std::max(1u, std::thread::hardware_concurrency()));
        ctx->workers = new malcs_workers(threads);
    }
    return ctx->workers;
}

static void ctx_release(malcs_ctx *c)
{
    delete c->workers;
    c->workers = nullptr;
    for (malcs_batch *b : c->spare_batches) delete b;
    c->spare_batches.clear();
    malcs_pool_trim(c);
    if (c->pack_stage) hipHostFree(c->pack_stage);
    if (c->stage) hipHostFree(c->stage);
    if (c->ev0) hipEventDestroy(c->ev0);
    if (c->ev1) hipEventDestroy(c->ev1);
    if (c->ev2) hipEventDestroy(c->ev2);
    if (c->copy_stream) hipStreamDestroy(c->copy_stream);
    if (c->stream) hipStreamDestroy(c->stream);
    delete c;
}

static int ctx_setup(malcs_ctx *c)
{
    MALCS_HIP(hipSetDevice(c->device));
    MALCS_HIP(hipGetDeviceProperties(&c->prop, c->device));
    if (strncmp(c->prop.gcnArchName, "gfx950", 6) != 0) {
        malcs_set_error("device %d is %s; libmalcs is built for gfx950 only", c->device, c->prop.gcnArchName);
        return MALCS_ERR_HIP;
    }
    MALCS_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    MALCS_HIP(hipEventCreate(&c->ev0));
    MALCS_HIP(hipEventCreate(&c->ev1));
    MALCS_HIP(hipEventCreate(&c->ev2));
    if (hipHostMalloc((void **)&c->stage, STAGE_BYTES, hipHostMallocDefault) == hipSuccess) c->stage_cap = STAGE_BYTES;
    else { c->stage = nullptr; (void)hipGetLastError(); }        // no staging: copies go straight from pageable memory
    return malcs_noalloc_setup(c);
}

extern "C" {

int malcs_ctx_create(int device, malcs_ctx **out)
{
    if (!out) { malcs_set_error("ctx is NULL"); return MALCS_ERR_ARG; }
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n == 0) {
        malcs_set_error("no HIP device available (%s); libmalcs has no CPU fallback",
                        e != hipSuccess ? hipGetErrorString(e) : "device count 0");
        return MALCS_ERR_HIP;
    }
    if (device < 0 || device >= n) { malcs_set_error("device %d out of range [0,%d)", device, n); return MALCS_ERR_ARG; }
    malcs_ctx *c = new (std::nothrow) malcs_ctx();
    if (!c) { malcs_set_error("out of host memory"); return MALCS_ERR_NOMEM; }
    c->device = device;
    const int rc = ctx_setup(c);
    if (rc) { ctx_release(c); return rc; }
    {
        std::lock_guard<std::mutex> lock(g_ctx_mutex);
        g_ctxs.push_back(c);
    }
    *out = c;
    return MALCS_OK;
}

int malcs_ctx_destroy(malcs_ctx *c)
{
    if (!c) return MALCS_OK;
    hipSetDevice(c->device);
    if (c->stream) malcs_sync(c->stream);
    {
        std::lock_guard<std::mutex> lock(g_ctx_mutex);
        for (size_t i = 0; i < g_ctxs.size(); i++) if (g_ctxs[i] == c) { g_ctxs.erase(g_ctxs.begin() + (std::ptrdiff_t)i); break; }
    }
    ctx_release(c);
    return MALCS_OK;
}

int malcs_ctx_sync(malcs_ctx *c)
{
    if (!c) { malcs_set_error("ctx is NULL"); return MALCS_ERR_ARG; }
    MALCS_HIP(hipSetDevice(c->device));
    MALCS_HIP(malcs_sync(c->stream));
    return MALCS_OK;
}

int malcs_ctx_device_info(malcs_ctx *c, char *name, int name_cap, int *cus, int *clock_mhz)
{
    if (!c) { malcs_set_error("ctx is NULL"); return MALCS_ERR_ARG; }
    if (name && name_cap > 0) snprintf(name, (size_t)name_cap, "%s (%s)", c->prop.name, c->prop.gcnArchName);
    if (cus) *cus = c->prop.multiProcessorCount;
    if (clock_mhz) *clock_mhz = c->prop.clockRate / 1000;
    return MALCS_OK;
}

static int planes_for(int n_symbols)
{
    if (n_symbols <= 4) return 2;
    if (n_symbols <= 8) return 3;
    if (n_symbols <= 16) return 4;
    return 8;
}

static void batch_release(malcs_batch *b)
{
    if (!b) return;
    hipSetDevice(b->ctx->device);
    malcs_diag_release(b);
    malcs_pool_free(b->ctx, b->d_planes);
    malcs_pool_free(b->ctx, b->d_pair_block);        // d_a_off ... d_order, d_dist are parts of it
    malcs_ctx *c = b->ctx;
    if (c->spare_batches.size() < 2 && b->n_pairs <= (1u << 22)) {    // keep the object: its host arrays have their pages
        b->d_planes = b->d_pair_block = b->d_a_off = b->d_a_len = b->d_b_off = b->d_b_len = b->d_order = nullptr;
        b->d_dist = nullptr;
        b->dist_buckets.clear(); b->band_pairs.clear();
        b->n_pairs = 0; b->full_cells = 0; b->arena_dwords = 0; b->max_len = 0;
        c->spare_batches.push_back(b);
        return;
    }
    delete b;
}

// Do the 2 n sequences of a batch lie apart in the symbol buffer?  Then none is shared and every one gets its own place
// in the arena without looking anything up.  Recognises the two usual layouts in one pass: a_0 b_0 a_1 b_1 ... and all
// a_i followed by all b_i (or the other way round), each in ascending order.
static bool sequences_apart(const uint64_t *a_off, const uint32_t *a_len, const uint64_t *b_off, const uint32_t *b_len, uint64_t n)
{
    bool interleaved = true, blocks = true;
    uint64_t end = 0, a_end = 0, b_end = 0, a_min = ~0ull, b_min = ~0ull;
    for (uint64_t i = 0; i < n && (interleaved || blocks); i++) {
        interleaved = interleaved && a_off[i] >= end && b_off[i] >= a_off[i] + a_len[i];
        end = b_off[i] + b_len[i];
        blocks = blocks && a_off[i] >= a_end && b_off[i] >= b_end;
        a_end = a_off[i] + a_len[i]; b_end = b_off[i] + b_len[i];
        a_min = std::min(a_min, a_off[i]); b_min = std::min(b_min, b_off[i]);
    }
    return interleaved || (blocks && (a_end <= b_min || b_end <= a_min));
}

// `syms`: the caller's symbols in host memory -- or, with `d_syms`, already in device memory (callers inside the library
// that put sequences together on the device: the relation matrix's union windows), then packed there by k_pack_planes
static int batch_build(malcs_ctx *ctx, const uint8_t *syms, const uint8_t *d_syms, uint64_t n_syms,
                       const uint64_t *a_off, const uint32_t *a_len, const uint64_t *b_off, const uint32_t *b_len,
                       uint64_t n_pairs, int n_symbols, malcs_batch *b)
{
    malcs_stage_trace trace("batch_create");
    b->ctx = ctx;
    b->n_pairs = n_pairs;
    b->n_planes = planes_for(n_symbols);
    const uint32_t NP = (uint32_t)b->n_planes;

    // ---- where every sequence goes in the arena ----------------------------------------------------------
    std::vector<uint64_t> &seq_byte_off = ctx->seq_byte_off;
    std::vector<uint32_t> &seq_len = ctx->seq_len, &seq_blk_off = ctx->seq_blk_off;
    uint64_t total_blocks = 0;
    bool placed = false;
    b->a_off.resize(n_pairs); b->a_len.resize(n_pairs); b->b_off.resize(n_pairs); b->b_len.resize(n_pairs);
    b->max_len = 0;
    const bool threaded = n_pairs >= 16384 && syms != nullptr;
    auto report_bad = [&]() -> int {                             // names the first sequence that is not inside the symbol buffer
        for (uint64_t i = 0; i < n_pairs; i++) {
            const bool bad_a = a_off[i] > n_syms || (uint64_t)a_len[i] > n_syms - a_off[i];
            const bool bad_b = b_off[i] > n_syms || (uint64_t)b_len[i] > n_syms - b_off[i];
            if (!bad_a && !bad_b) continue;
            malcs_set_error("sequence [%llu,+%u) outside the symbol buffer of %llu bytes", (unsigned long long)(bad_a ? a_off[i] : b_off[i]),
                            bad_a ? a_len[i] : b_len[i], (unsigned long long)n_syms);
            return MALCS_ERR_ARG;
        }
        return MALCS_OK;
    };
    if (!threaded) {                                             // (large batches: checked by the threaded pass below)
        uint64_t bad = 0;                                        // every sequence inside the symbol buffer?
        for (uint64_t i = 0; i < n_pairs; i++)
            bad |= (uint64_t)(a_off[i] > n_syms) | (uint64_t)((uint64_t)a_len[i] > n_syms - a_off[i]) |
                   (uint64_t)(b_off[i] > n_syms) | (uint64_t)((uint64_t)b_len[i] > n_syms - b_off[i]);
        if (bad) return report_bad();
    }
    if (threaded) {
        // the same three passes (inside the buffer? apart? places) over stretches of pairs, on the context's threads; a
        // batch that shares sequences, or has a bad one, goes through the serial code below for its table / its message
        malcs_workers *wk = malcs_ctx_workers(ctx);
        const int T = wk->size();
        struct part { uint64_t blocks = 0, a_min = ~0ull, b_min = ~0ull; uint32_t max_len = 0; bool bad = false, interleaved = true, ordered = true; };
        std::vector<part> parts((size_t)T);
        auto range = [&](int k, uint64_t &lo, uint64_t &hi) { lo = n_pairs * (uint64_t)k / (uint64_t)T; hi = n_pairs * (uint64_t)(k + 1) / (uint64_t)T; };
        wk->run(T, [&](int k) {
            uint64_t lo, hi; range(k, lo, hi);
            part pt;
            uint64_t bad = 0;
            uint64_t end = lo ? b_off[lo - 1] + b_len[lo - 1] : 0, a_end = lo ? a_off[lo - 1] + a_len[lo - 1] : 0, b_end = end;
            for (uint64_t i = lo; i < hi; i++) {
                bad |= (uint64_t)(a_off[i] > n_syms) | (uint64_t)((uint64_t)a_len[i] > n_syms - a_off[i]) |
                       (uint64_t)(b_off[i] > n_syms) | (uint64_t)((uint64_t)b_len[i] > n_syms - b_off[i]);
                pt.interleaved = pt.interleaved && a_off[i] >= end && b_off[i] >= a_off[i] + a_len[i];
                end = b_off[i] + b_len[i];
                pt.ordered = pt.ordered && a_off[i] >= a_end && b_off[i] >= b_end;
                a_end = a_off[i] + a_len[i]; b_end = end;
                pt.a_min = std::min(pt.a_min, a_off[i]); pt.b_min = std::min(pt.b_min, b_off[i]);
                pt.blocks += (a_len[i] + 31u) / 32u + (b_len[i] + 31u) / 32u;
                pt.max_len = std::max(pt.max_len, std::max(a_len[i], b_len[i]));
            }
            pt.bad = bad != 0;
            parts[(size_t)k] = pt;
        });
        bool bad = false, interleaved = true, ordered = true;
        uint64_t a_min = ~0ull, b_min = ~0ull, blocks = 0;
        for (const part &pt : parts) {
            bad = bad || pt.bad; interleaved = interleaved && pt.interleaved; ordered = ordered && pt.ordered;
            a_min = std::min(a_min, pt.a_min); b_min = std::min(b_min, pt.b_min); blocks += pt.blocks;
        }
        if (bad) return report_bad();
        const uint64_t a_end = a_off[n_pairs - 1] + a_len[n_pairs - 1], b_end = b_off[n_pairs - 1] + b_len[n_pairs - 1];
        const bool apart = (interleaved || (ordered && (a_end <= b_min || b_end <= a_min)));
        if (apart && blocks * NP >= 0xffffffffull) { malcs_set_error("batch needs more than 16 GiB of packed planes; split it"); return MALCS_ERR_TOO_LONG; }
        if (apart) {
            seq_byte_off.resize((size_t)n_pairs * 2); seq_len.resize((size_t)n_pairs * 2); seq_blk_off.resize((size_t)n_pairs * 2);
            uint64_t acc = 0;
            for (part &pt : parts) { const uint64_t mine = pt.blocks; pt.blocks = acc; acc += mine; b->max_len = std::max(b->max_len, pt.max_len); }
            wk->run(T, [&](int k) {
                uint64_t lo, hi; range(k, lo, hi);
                uint64_t at = parts[(size_t)k].blocks;
                for (uint64_t i = lo; i < hi; i++) {
                    seq_byte_off[2 * i] = a_off[i]; seq_len[2 * i] = a_len[i]; seq_blk_off[2 * i] = (uint32_t)at;
                    b->a_off[i] = (uint32_t)at * NP; b->a_len[i] = a_len[i];
                    at += (a_len[i] + 31u) / 32u;
                    seq_byte_off[2 * i + 1] = b_off[i]; seq_len[2 * i + 1] = b_len[i]; seq_blk_off[2 * i + 1] = (uint32_t)at;
                    b->b_off[i] = (uint32_t)at * NP; b->b_len[i] = b_len[i];
                    at += (b_len[i] + 31u) / 32u;
                }
            });
            total_blocks = acc;
            placed = true;
        }
    }
    if (!placed) { seq_byte_off.clear(); seq_len.clear(); seq_blk_off.clear(); }   // (placed: every entry was written; no refill of 3 MB)
    if (placed) {
    } else if (sequences_apart(a_off, a_len, b_off, b_len, n_pairs)) {
        seq_byte_off.resize((size_t)n_pairs * 2); seq_len.resize((size_t)n_pairs * 2); seq_blk_off.resize((size_t)n_pairs * 2);
        for (uint64_t i = 0; i < n_pairs; i++) {
            seq_byte_off[2 * i] = a_off[i]; seq_len[2 * i] = a_len[i]; seq_blk_off[2 * i] = (uint32_t)total_blocks;
            b->a_off[i] = (uint32_t)total_blocks * NP; b->a_len[i] = a_len[i];
            total_blocks += (a_len[i] + 31u) / 32u;
            seq_byte_off[2 * i + 1] = b_off[i]; seq_len[2 * i + 1] = b_len[i]; seq_blk_off[2 * i + 1] = (uint32_t)total_blocks;
            b->b_off[i] = (uint32_t)total_blocks * NP; b->b_len[i] = b_len[i];
            total_blocks += (b_len[i] + 31u) / 32u;
            b->max_len = std::max(b->max_len, std::max(a_len[i], b_len[i]));
            if (total_blocks * NP >= 0xffffffffull) { malcs_set_error("batch needs more than 16 GiB of packed planes; split it"); return MALCS_ERR_TOO_LONG; }
        }
    } else {
        // pairs share sequences (a relation-matrix tile re-uses every window many times): distinct ones only.
        // Open addressing over (offset, length); the table holds sequence ids, empty slots are ~0.
        size_t table_size = 16;
        while (table_size < (size_t)n_pairs * 4 + 4) table_size <<= 1;
        std::vector<uint32_t> table(table_size, 0xffffffffu);
        seq_byte_off.reserve((size_t)n_pairs * 2); seq_len.reserve((size_t)n_pairs * 2); seq_blk_off.reserve((size_t)n_pairs * 2);
        auto intern = [&](uint64_t off, uint32_t len) -> uint32_t {
            size_t slot = (size_t)((off * 0x9e3779b97f4a7c15ull) ^ ((uint64_t)len * 0xc2b2ae3d27d4eb4full)) >> 20 & (table_size - 1);
            for (; table[slot] != 0xffffffffu; slot = (slot + 1) & (table_size - 1)) {
                const uint32_t have = table[slot];
                if (seq_byte_off[have] == off && seq_len[have] == len) return have;
            }
            table[slot] = (uint32_t)seq_len.size();
            seq_byte_off.push_back(off);
            seq_len.push_back(len);
            seq_blk_off.push_back((uint32_t)total_blocks);
            total_blocks += (len + 31u) / 32u;
            return (uint32_t)seq_len.size() - 1u;
        };
        for (uint64_t i = 0; i < n_pairs; i++) {
            const uint32_t ia = intern(a_off[i], a_len[i]), ib = intern(b_off[i], b_len[i]);
            if (total_blocks * NP >= 0xffffffffull) { malcs_set_error("batch needs more than 16 GiB of packed planes; split it"); return MALCS_ERR_TOO_LONG; }
            b->a_off[i] = seq_blk_off[ia] * NP; b->a_len[i] = a_len[i];
            b->b_off[i] = seq_blk_off[ib] * NP; b->b_len[i] = b_len[i];
            b->max_len = std::max(b->max_len, std::max(a_len[i], b_len[i]));
        }
    }
    const uint32_t n_seqs = (uint32_t)seq_len.size();
    b->arena_dwords = total_blocks * NP;
    trace.lap("places in the arena");

    MALCS_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    if (n_pairs) {
        // one block for everything per pair: a_off, a_len, b_off, b_len (copied below), order (malcs_dist_plan), dist
        MALCS_HIP(malcs_pool_alloc(ctx, (void **)&b->d_planes, (b->arena_dwords + 4) * sizeof(uint32_t)));
        const size_t np = ((size_t)n_pairs + 63) & ~(size_t)63;
        MALCS_HIP(malcs_pool_alloc(ctx, (void **)&b->d_pair_block, 6 * np * sizeof(uint32_t)));
        b->d_a_off = b->d_pair_block; b->d_a_len = b->d_pair_block + np; b->d_b_off = b->d_pair_block + 2 * np;
        b->d_b_len = b->d_pair_block + 3 * np; b->d_order = b->d_pair_block + 4 * np;
        b->d_dist = reinterpret_cast<int32_t *>(b->d_pair_block + 5 * np);
        MALCS_HIP(malcs_h2d(st, b->d_a_off, b->a_off.data(), n_pairs * 4));
        MALCS_HIP(malcs_h2d(st, b->d_a_len, b->a_len.data(), n_pairs * 4));
        MALCS_HIP(malcs_h2d(st, b->d_b_off, b->b_off.data(), n_pairs * 4));
        MALCS_HIP(malcs_h2d(st, b->d_b_len, b->b_len.data(), n_pairs * 4));
        if (total_blocks && d_syms) {
            pool_scoped<uint64_t> d_sbo(ctx);
            pool_scoped<uint32_t> d_sl(ctx), d_sko(ctx);
            MALCS_HIP(d_sbo.alloc(n_seqs)); MALCS_HIP(d_sl.alloc(n_seqs)); MALCS_HIP(d_sko.alloc(n_seqs));
            MALCS_HIP(malcs_h2d(st, d_sbo.p, seq_byte_off.data(), (size_t)n_seqs * 8));
            MALCS_HIP(malcs_h2d(st, d_sl.p, seq_len.data(), (size_t)n_seqs * 4));
            MALCS_HIP(malcs_h2d(st, d_sko.p, seq_blk_off.data(), (size_t)n_seqs * 4));
            const int rc = malcs_pack_launch(ctx, d_syms, d_sbo.p, d_sl.p, d_sko.p, n_seqs, (uint32_t)total_blocks, b->n_planes, b->d_planes);
            if (rc) return rc;
            if ((size_t)n_seqs * 8 > (1u << 20)) MALCS_HIP(malcs_sync(st));     // copies beyond the staging buffer read the vectors directly
        } else if (total_blocks) {
            // the planes are packed here, on the host's cores, into pinned memory, and cross PCIe packed; every stretch
            // the packers finish is queued for its copy at once
            const size_t arena_bytes = (size_t)b->arena_dwords * sizeof(uint32_t);
            if (ctx->pack_stage_cap < arena_bytes) {
                MALCS_HIP(malcs_sync(st));                       // an earlier batch's copies may still read the old buffer
                if (ctx->pack_stage) { (void)hipHostFree(ctx->pack_stage); ctx->pack_stage = nullptr; ctx->pack_stage_cap = 0; }
                const size_t cap = arena_bytes + arena_bytes / 4 + (1u << 20);
                if (hipHostMalloc((void **)&ctx->pack_stage, cap, hipHostMallocDefault) != hipSuccess) {
                    (void)hipGetLastError();
                    malcs_set_error("no pinned host memory for %zu bytes of packed planes", cap);
                    return MALCS_ERR_NOMEM;
                }
                ctx->pack_stage_cap = cap;
            }                                                    // (the buffer is idle: batch creation ends with the stream drained)
            malcs_pack_job job{syms, seq_byte_off.data(), seq_len.data(), seq_blk_off.data(), n_seqs, b->n_planes,
                               reinterpret_cast<uint32_t *>(ctx->pack_stage)};
            hipError_t copy_error = hipSuccess;
            uint32_t codes = 0;
            int threads = 16;
            if (const char *e = getenv("MALCS_PACK_THREADS")) threads = std::max(1, atoi(e));
            threads = std::min<int>(threads, (int)std::max(1u, std::thread::hardware_concurrency()));
            const int rc = malcs_host_pack(job, threads, [&](size_t first, size_t count) -> int {
                copy_error = hipMemcpyAsync(b->d_planes + first, job.out + first, count * sizeof(uint32_t), hipMemcpyHostToDevice, st);
                return copy_error == hipSuccess ? 0 : 1;
            }, &codes, threads > 1 && n_pairs >= 16384 ? malcs_ctx_workers(ctx) : nullptr);
            if (rc) { malcs_set_error("copy of the packed planes failed: %s", hipGetErrorString(copy_error)); return MALCS_ERR_HIP; }
            if (codes >> NP) {
                malcs_set_error("symbol code %u and n_symbols = %d: the batch has codes beyond the %u bit planes n_symbols selects", codes, n_symbols, NP);
                return MALCS_ERR_ARG;
            }
            if (n_symbols < (1 << NP) && codes >= (uint32_t)n_symbols) {
                // the OR of all codes is an upper bound of every one; only when it is itself out of range can a code be
                bool bad = false;
                for (uint32_t s = 0; s < n_seqs && !bad; s++)
                    for (uint32_t j = 0; j < seq_len[s]; j++) if (syms[seq_byte_off[s] + j] >= n_symbols) { bad = true; break; }
                if (bad) { malcs_set_error("a symbol code is not below n_symbols = %d", n_symbols); return MALCS_ERR_ARG; }
            }
        }
    }
    trace.lap("allocations, planes packed on the host, copies queued");
    const int rc = malcs_dist_plan(b);
    trace.lap("distance plan");
    return rc;
}

int malcs_batch_create(malcs_ctx *ctx, const uint8_t *syms, uint64_t n_syms,
                       const uint64_t *a_off, const uint32_t *a_len, const uint64_t *b_off, const uint32_t *b_len,
                       uint64_t n_pairs, int n_symbols, malcs_batch **out)
{
    if (!out) { malcs_set_error("batch is NULL"); return MALCS_ERR_ARG; }
    *out = nullptr;
    if (!ctx) { malcs_set_error("ctx is NULL"); return MALCS_ERR_ARG; }
    if (n_pairs && (!a_off || !a_len || !b_off || !b_len)) { malcs_set_error("offset/length arrays are NULL"); return MALCS_ERR_ARG; }
    if (n_syms && !syms) { malcs_set_error("syms is NULL"); return MALCS_ERR_ARG; }
    if (n_symbols < 1 || n_symbols > 256) { malcs_set_error("n_symbols %d outside [1,256]", n_symbols); return MALCS_ERR_ARG; }
    if (n_pairs >= 0x7fffffffull) { malcs_set_error("too many pairs in one batch"); return MALCS_ERR_ARG; }
    malcs_batch *b = nullptr;
    if (!ctx->spare_batches.empty()) { b = ctx->spare_batches.back(); ctx->spare_batches.pop_back(); }
    else b = new (std::nothrow) malcs_batch();
    if (!b) { malcs_set_error("out of host memory"); return MALCS_ERR_NOMEM; }
    int rc = batch_build(ctx, syms, nullptr, n_syms, a_off, a_len, b_off, b_len, n_pairs, n_symbols, b);
    if (rc) {
        (void)hipStreamSynchronize(ctx->stream);                 // copies out of the context's pinned buffers may be in flight
        batch_release(b);
        return rc;
    }
    *out = b;
    return MALCS_OK;
}

int malcs_batch_create_packed(malcs_ctx *ctx, const uint32_t *planes, uint64_t n_dwords, int planes_on_device,
                              const uint32_t *a_off, const uint32_t *a_len, const uint32_t *b_off, const uint32_t *b_len,
                              uint64_t n_pairs, int n_planes, malcs_batch **out)
{
    if (!out) { malcs_set_error("batch is NULL"); return MALCS_ERR_ARG; }
    *out = nullptr;
    if (!ctx) { malcs_set_error("ctx is NULL"); return MALCS_ERR_ARG; }
    if (n_pairs && (!a_off || !a_len || !b_off || !b_len)) { malcs_set_error("offset/length arrays are NULL"); return MALCS_ERR_ARG; }
    if (n_dwords && !planes) { malcs_set_error("planes is NULL"); return MALCS_ERR_ARG; }
    if (n_planes != 2 && n_planes != 3 && n_planes != 4 && n_planes != 8) { malcs_set_error("n_planes %d is not 2, 3, 4 or 8", n_planes); return MALCS_ERR_ARG; }
    if (n_pairs >= 0x7fffffffull) { malcs_set_error("too many pairs in one batch"); return MALCS_ERR_ARG; }
    if (n_dwords >= 0xffffffffull) { malcs_set_error("batch needs more than 16 GiB of packed planes; split it"); return MALCS_ERR_TOO_LONG; }
    const uint32_t NP = (uint32_t)n_planes;
    malcs_stage_trace trace("batch_create_packed");
    uint32_t max_len = 0;
    {
        // every run of blocks inside the arena and at a block?  One pass without branches; the pair at fault is looked
        // for only when there is one
        uint64_t bad = 0;
        uint32_t longest = 0;
        auto run_bad = [&](uint64_t off, uint64_t len) -> uint64_t {
            return (uint64_t)(off % NP != 0) | (uint64_t)(off + (len + 31u) / 32u * NP > n_dwords);
        };
        if (NP == 3) {
            for (uint64_t i = 0; i < n_pairs; i++) {
                bad |= run_bad(a_off[i], a_len[i]) | run_bad(b_off[i], b_len[i]);
                longest = std::max(longest, std::max(a_len[i], b_len[i]));
            }
        } else {
            const uint32_t low = NP - 1u, shift = NP == 2 ? 1u : NP == 4 ? 2u : 3u;
            for (uint64_t i = 0; i < n_pairs; i++) {
                const uint64_t ea = (uint64_t)a_off[i] + ((((uint64_t)a_len[i] + 31u) >> 5) << shift);
                const uint64_t eb = (uint64_t)b_off[i] + ((((uint64_t)b_len[i] + 31u) >> 5) << shift);
                bad |= (uint64_t)((a_off[i] | b_off[i]) & low) | (uint64_t)(ea > n_dwords) | (uint64_t)(eb > n_dwords);
                longest = std::max(longest, std::max(a_len[i], b_len[i]));
            }
        }
        max_len = longest;
        for (uint64_t i = 0; bad && i < n_pairs; i++)
            for (int side = 0; side < 2; side++) {
                const uint64_t off = side ? b_off[i] : a_off[i], len = side ? b_len[i] : a_len[i];
                if (!run_bad(off, len)) continue;
                malcs_set_error("pair %llu: sequence of %llu symbols at dword %llu is not a run of whole blocks inside the arena of %llu dwords",
                                (unsigned long long)i, (unsigned long long)len, (unsigned long long)off, (unsigned long long)n_dwords);
                return MALCS_ERR_ARG;
            }
    }
    malcs_batch *b = nullptr;
    if (!ctx->spare_batches.empty()) { b = ctx->spare_batches.back(); ctx->spare_batches.pop_back(); }
    else b = new (std::nothrow) malcs_batch();
    if (!b) { malcs_set_error("out of host memory"); return MALCS_ERR_NOMEM; }
    trace.lap("tables checked");
    auto build = [&]() -> int {
        b->ctx = ctx; b->n_pairs = n_pairs; b->n_planes = n_planes; b->max_len = max_len; b->arena_dwords = n_dwords;
        b->a_off.assign(a_off, a_off + n_pairs); b->a_len.assign(a_len, a_len + n_pairs);
        b->b_off.assign(b_off, b_off + n_pairs); b->b_len.assign(b_len, b_len + n_pairs);
        MALCS_HIP(hipSetDevice(ctx->device));
        hipStream_t st = ctx->stream;
        if (n_pairs) {
            MALCS_HIP(malcs_pool_alloc(ctx, (void **)&b->d_planes, (b->arena_dwords + 4) * sizeof(uint32_t)));
            const size_t np = ((size_t)n_pairs + 63) & ~(size_t)63;
            MALCS_HIP(malcs_pool_alloc(ctx, (void **)&b->d_pair_block, 6 * np * sizeof(uint32_t)));
            b->d_a_off = b->d_pair_block; b->d_a_len = b->d_pair_block + np; b->d_b_off = b->d_pair_block + 2 * np;
            b->d_b_len = b->d_pair_block + 3 * np; b->d_order = b->d_pair_block + 4 * np;
            b->d_dist = reinterpret_cast<int32_t *>(b->d_pair_block + 5 * np);
            if (n_dwords)                                        // the library's own copy: the caller's arena is free again on return
                MALCS_HIP(hipMemcpyAsync(b->d_planes, planes, n_dwords * sizeof(uint32_t),
                                         planes_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, st));
            MALCS_HIP(hipMemsetAsync(b->d_planes + n_dwords, 0, 4 * sizeof(uint32_t), st));
            MALCS_HIP(malcs_h2d(st, b->d_a_off, b->a_off.data(), n_pairs * 4));
            MALCS_HIP(malcs_h2d(st, b->d_a_len, b->a_len.data(), n_pairs * 4));
            MALCS_HIP(malcs_h2d(st, b->d_b_off, b->b_off.data(), n_pairs * 4));
            MALCS_HIP(malcs_h2d(st, b->d_b_len, b->b_len.data(), n_pairs * 4));
            const int rc = malcs_clear_tails_launch(ctx, b->d_planes, b->d_a_off, b->d_a_len, b->d_b_off, b->d_b_len, (uint32_t)n_pairs, n_planes);
            if (rc) return rc;
            if (!planes_on_device) MALCS_HIP(malcs_sync(st));    // pageable host planes: the copy must be over before the caller has them back
        }
        trace.lap("allocations, copies queued");
        const int rc = malcs_dist_plan(b);
        trace.lap("distance plan");
        return rc;
    };
    const int rc = build();
    if (rc) { (void)hipStreamSynchronize(ctx->stream); batch_release(b); return rc; }
    *out = b;
    return MALCS_OK;
}

}  // extern "C"

int malcs_batch_create_device(malcs_ctx *ctx, const uint8_t *d_syms, uint64_t n_syms,
                              const uint64_t *a_off, const uint32_t *a_len, const uint64_t *b_off, const uint32_t *b_len,
                              uint64_t n_pairs, int n_symbols, malcs_batch **out)
{
    *out = nullptr;
    malcs_batch *b = nullptr;
    if (!ctx->spare_batches.empty()) { b = ctx->spare_batches.back(); ctx->spare_batches.pop_back(); }
    else b = new (std::nothrow) malcs_batch();
    if (!b) { malcs_set_error("out of host memory"); return MALCS_ERR_NOMEM; }
    const int rc = batch_build(ctx, nullptr, d_syms, n_syms, a_off, a_len, b_off, b_len, n_pairs, n_symbols, b);
    if (rc) { (void)hipStreamSynchronize(ctx->stream); batch_release(b); return rc; }
    *out = b;
    return MALCS_OK;
}

extern "C" {

int malcs_batch_destroy(malcs_batch *b)
{
    if (!b) return MALCS_OK;
    hipSetDevice(b->ctx->device);
    malcs_sync(b->ctx->stream);
    batch_release(b);
    return MALCS_OK;
}

uint64_t malcs_batch_size(const malcs_batch *b) { return b ? b->n_pairs : 0; }

int malcs_batch_replan(malcs_batch *b)
{
    if (!b) { malcs_set_error("batch is NULL"); return MALCS_ERR_ARG; }
    MALCS_HIP(hipSetDevice(b->ctx->device));
    return malcs_dist_replan(b);
}

int malcs_edit_distance_launch(malcs_batch *b)
{
    if (!b) { malcs_set_error("batch is NULL"); return MALCS_ERR_ARG; }
    MALCS_HIP(hipSetDevice(b->ctx->device));
    return malcs_dist_launch(b);
}

int malcs_edit_distance_fetch(malcs_batch *b, int32_t *dist)
{
    if (!b || (!dist && b->n_pairs)) { malcs_set_error("NULL argument"); return MALCS_ERR_ARG; }
    MALCS_HIP(hipSetDevice(b->ctx->device));
    if (b->n_pairs)
        MALCS_HIP(malcs_d2h(b->ctx->stream, dist, b->d_dist, b->n_pairs * sizeof(int32_t)));
    MALCS_HIP(malcs_sync(b->ctx->stream));
    return MALCS_OK;
}

int malcs_edit_distance_time(malcs_batch *b, int iters, float *ms_total)
{
    if (!b || !ms_total || iters < 1) { malcs_set_error("bad argument"); return MALCS_ERR_ARG; }
    malcs_ctx *c = b->ctx;
    MALCS_HIP(hipSetDevice(c->device));
    MALCS_HIP(hipEventRecord(c->ev0, c->stream));
    for (int i = 0; i < iters; i++) {
        int rc = malcs_dist_launch(b);
        if (rc) return rc;
    }
    MALCS_HIP(hipEventRecord(c->ev1, c->stream));
    MALCS_HIP(hipEventSynchronize(c->ev1));
    MALCS_HIP(hipEventElapsedTime(ms_total, c->ev0, c->ev1));
    return MALCS_OK;
}

int malcs_edit_distance_time_kernel(malcs_batch *b, int iters, float *ms_total)
{
    if (!b || !ms_total || iters < 1) { malcs_set_error("bad argument"); return MALCS_ERR_ARG; }
    if (!b->diag.nlw) return malcs_edit_distance_time(b, iters, ms_total);
    malcs_ctx *c = b->ctx;
    MALCS_HIP(hipSetDevice(c->device));
    MALCS_HIP(hipEventRecord(c->ev0, c->stream));
    for (int i = 0; i < iters; i++) {
        int rc = malcs_diag_launch(b, true);       // every launch starts from zeroed retry counters, as in a real step
        if (rc) return rc;
    }
    MALCS_HIP(hipEventRecord(c->ev1, c->stream));
    MALCS_HIP(hipEventSynchronize(c->ev1));
    MALCS_HIP(hipEventElapsedTime(ms_total, c->ev0, c->ev1));
    return MALCS_OK;
}

int malcs_edit_distance_kernel_durations(malcs_batch *b, int iters, float *ms_each)
{
    if (!b || !ms_each || iters < 1) { malcs_set_error("bad argument"); return MALCS_ERR_ARG; }
    if (!b->diag.nlw) { malcs_set_error("the batch has no window class: use malcs_edit_distance_time_kernel"); return MALCS_ERR_ARG; }
    malcs_ctx *c = b->ctx;
    MALCS_HIP(hipSetDevice(c->device));
    std::vector<hipEvent_t> ev((size_t)2 * iters, nullptr);
    int rc = MALCS_OK;
    for (hipEvent_t &e : ev)
        if (hipEventCreate(&e) != hipSuccess) { (void)hipGetLastError(); malcs_set_error("hipEventCreate failed"); rc = MALCS_ERR_HIP; break; }
    for (int i = 0; i < iters && rc == MALCS_OK; i++) rc = malcs_diag_launch(b, true, ev[2 * (size_t)i], ev[2 * (size_t)i + 1]);
    if (hipStreamSynchronize(c->stream) != hipSuccess && rc == MALCS_OK) { malcs_set_error("synchronisation failed"); rc = MALCS_ERR_HIP; }
    for (int i = 0; i < iters && rc == MALCS_OK; i++)
        if (hipEventElapsedTime(&ms_each[i], ev[2 * (size_t)i], ev[2 * (size_t)i + 1]) != hipSuccess) { (void)hipGetLastError(); malcs_set_error("hipEventElapsedTime failed"); rc = MALCS_ERR_HIP; }
    for (hipEvent_t e : ev) if (e) (void)hipEventDestroy(e);
    return rc;
}

int malcs_edit_distance_step_durations(malcs_batch *b, int iters, float *ms_each)
{
    if (!b || !ms_each || iters < 1) { malcs_set_error("bad argument"); return MALCS_ERR_ARG; }
    if (!b->diag.nlw) { malcs_set_error("the batch has no window class"); return MALCS_ERR_ARG; }
    malcs_ctx *c = b->ctx;
    MALCS_HIP(hipSetDevice(c->device));
    std::vector<hipEvent_t> ev((size_t)2 * iters, nullptr);
    int rc = MALCS_OK;
    for (hipEvent_t &e : ev)
        if (hipEventCreate(&e) != hipSuccess) { (void)hipGetLastError(); malcs_set_error("hipEventCreate failed"); rc = MALCS_ERR_HIP; break; }
    for (int i = 0; i < iters && rc == MALCS_OK; i++) {
        rc = malcs_dist_replan(b);
        if (rc == MALCS_OK) rc = malcs_dist_launch(b, ev[2 * (size_t)i], ev[2 * (size_t)i + 1]);
    }
    if (hipStreamSynchronize(c->stream) != hipSuccess && rc == MALCS_OK) { malcs_set_error("synchronisation failed"); rc = MALCS_ERR_HIP; }
    for (int i = 0; i < iters && rc == MALCS_OK; i++)
        if (hipEventElapsedTime(&ms_each[i], ev[2 * (size_t)i], ev[2 * (size_t)i + 1]) != hipSuccess) { (void)hipGetLastError(); malcs_set_error("hipEventElapsedTime failed"); rc = MALCS_ERR_HIP; }
    for (hipEvent_t e : ev) if (e) (void)hipEventDestroy(e);
    return rc;
}

int malcs_batch_plan_info(malcs_batch *b, uint64_t *info)
{
    if (!b || !info) { malcs_set_error("NULL argument"); return MALCS_ERR_ARG; }
    MALCS_HIP(hipSetDevice(b->ctx->device));
    uint32_t st[PS_FIELDS];
    const int rc = malcs_diag_stats(b, st);                      // what the plan decided stays on the device until somebody asks
    if (rc) return rc;
    const diag_plan &d = b->diag;
    const bool all_half = d.nlw && st[PS_NHALF] == d.n_tiles;
    const uint64_t cols0 = (uint64_t)st[PS_COLS0_HI] << 32 | st[PS_COLS0_LO], cols1 = (uint64_t)st[PS_COLS1_HI] << 32 | st[PS_COLS1_LO];
    // cells as laid out: rows of a tile's mode x 32 columns x its blocks x its pairs (the last tile may hold fewer than 64)
    const uint64_t rows3 = 32ull * (uint64_t)d.nlw, missing = d.n_pairs % 64u ? 64u - d.n_pairs % 64u : 0u;
    uint64_t cells = 0;
    if (d.nlw) cells = 32ull * 64ull * (64ull * st[PS_BLOCKS_HALF] + rows3 * (st[PS_BLOCKS] - st[PS_BLOCKS_HALF]))
                       - 32ull * missing * st[PS_IT_LAST] * (st[PS_LAST_HALF] ? 64ull : rows3);
    info[0] = d.n_pairs; info[1] = (uint64_t)d.nlw; info[2] = (uint64_t)(all_half ? 48 : d.T);
    info[3] = cells; info[4] = b->full_cells; info[5] = b->dist_buckets.size();
    info[6] = d.n_tiles; info[7] = st[PS_NHALF];
    info[8] = st[PS_NPIECES]; info[9] = st[PS_NCUT]; info[10] = d.n_launch;
    // the cells between the common ends, as they are (no growth to whole blocks or to the tile's longest): columns of
    // either group x the rows its mode computes
    info[11] = cols0 * (st[PS_HALFMODE] ? 64ull : rows3) + cols1 * rows3;
    info[12] = st[PS_SAMPLE_MEAN_X16]; info[13] = st[PS_SAMPLE_SD_X16]; info[14] = st[PS_WORST]; info[15] = st[PS_BALANCED];
    return MALCS_OK;
}

int malcs_batch_plan_dump(malcs_batch *b, int what, uint32_t *out, uint64_t cap, uint64_t *n_out)
{
    if (!b || !n_out) { malcs_set_error("NULL argument"); return MALCS_ERR_ARG; }
    MALCS_HIP(hipSetDevice(b->ctx->device));
    return malcs_diag_dump(b, what, out, cap, n_out);
}

int malcs_edit_distance_batch(malcs_ctx *ctx, const uint8_t *syms, uint64_t n_syms,
                              const uint64_t *a_off, const uint32_t *a_len, const uint64_t *b_off, const uint32_t *b_len,
                              uint64_t n_pairs, int n_symbols, int32_t *dist)
{
    malcs_batch *b = nullptr;
    int rc = malcs_batch_create(ctx, syms, n_syms, a_off, a_len, b_off, b_len, n_pairs, n_symbols, &b);
    if (rc) return rc;
    rc = malcs_edit_distance_launch(b);
    if (!rc) rc = malcs_edit_distance_fetch(b, dist);
    malcs_batch_destroy(b);
    return rc;
}

}  // extern "C"
