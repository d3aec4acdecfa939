// malcs_noalloc.hip -- the one entry point that allocates nothing: malcs_edit_distance_noalloc.
//
// No counterpart in the reference.  It is for callers that cannot call hipMalloc (code under HIP graph capture,
// tight serving loops, harnesses that forbid allocation): the distances of edit_distance(a, b)
// (algebra/lcs/edit_distance.py:21-53) for pairs whose symbols, offsets and lengths already sit in device memory,
// written to device memory, with every intermediate -- the packed bit planes, the per-sequence and per-pair tables,
// the work lists of the launch classes and their counters -- in ONE file-scope __device__ array (`scratch` below;
// every GPU the library runs on holds its own copy until the process ends).  The kernels that do the work are the
// ones malcs_edit_distance_launch uses for everything outside the window class (k_llcs_bitpar_list, malcs_dist.hip);
// what is new here is a plan that runs on the device: a scan of the block counts, the pack pass, and a classify pass
// that deals pair ids to per-class lists through atomic counters.
//
// Host code never uses the array's name as an address: malcs_noalloc_setup (called by malcs_ctx_create with the
// context's device current) asks hipGetSymbolAddress once per context and keeps the pointer in the context; kernels
// take that pointer as an argument.  All calls on one GPU share the array: it keeps the last call's contents (the
// counters are zeroed by the scan kernel of every call), and calls that overlap on one GPU race.
#include "malcs_internal.hpp"
#include "malcs_device.hpp"
#include <algorithm>

namespace {

constexpr uint32_t MAXP = MALCS_NOALLOC_MAX_PAIRS;                 // pairs per call
constexpr uint32_t MAXLEN = MALCS_NOALLOC_MAX_LEN;                 // shorter sequence of a pair: 64 lanes x 4 words
constexpr uint32_t N_CLASSES = 9;                                  // (W, L): (1,1) (2,1) (4,1) (4,2) (4,4) ... (4,64)
constexpr uint32_t PLANE_DWORDS = MALCS_NOALLOC_PLANE_DWORDS;      // packed arena
// dword offsets of the parts of the scratch array
constexpr size_t AT_PLANES = 0;
constexpr size_t AT_SEQ_BYTE_OFF = AT_PLANES + PLANE_DWORDS;       // 2 MAXP x uint64
constexpr size_t AT_SEQ_LEN = AT_SEQ_BYTE_OFF + 4u * MAXP;         // 2 MAXP; ~0 = the sequence cannot be handled
constexpr size_t AT_SEQ_BLK_OFF = AT_SEQ_LEN + 2u * MAXP;          // 2 MAXP
constexpr size_t AT_PAIR = AT_SEQ_BLK_OFF + 2u * MAXP;             // a_off, a_len, b_off, b_len: 4 x MAXP
constexpr size_t AT_LISTS = AT_PAIR + 4u * MAXP;                   // N_CLASSES x MAXP pair ids
constexpr size_t AT_COUNTERS = AT_LISTS + (size_t)N_CLASSES * MAXP;// [0, N_CLASSES): list lengths; [16]: blocks in the arena
constexpr size_t SCRATCH_DWORDS = AT_COUNTERS + 64;

__device__ __align__(256) uint32_t scratch[SCRATCH_DWORDS];

// Sequence s = 2 i + {0: a, 1: b} of pair i.  One workgroup: block counts, validity, exclusive scan.
__global__ void __launch_bounds__(1024)
k_na_scan(const uint64_t *__restrict__ a_off, const uint32_t *__restrict__ a_len,
          const uint64_t *__restrict__ b_off, const uint32_t *__restrict__ b_len,
          uint32_t n_pairs, uint64_t n_syms, uint32_t cap_blocks, uint32_t *__restrict__ sc)
{
    __shared__ uint32_t part[1024];
    const uint32_t t = threadIdx.x, n_seqs = 2u * n_pairs;
    const uint32_t chunk = (n_seqs + 1023u) / 1024u, s0 = t * chunk, s1 = min(n_seqs, s0 + chunk);
    uint64_t *seq_byte_off = reinterpret_cast<uint64_t *>(sc + AT_SEQ_BYTE_OFF);
    uint32_t *seq_len = sc + AT_SEQ_LEN, *seq_blk_off = sc + AT_SEQ_BLK_OFF;
    if (t < 64) sc[AT_COUNTERS + t] = 0u;
    auto blocks_of = [&](uint32_t s, uint64_t &off, uint32_t &len) -> uint32_t {
        const uint32_t i = s >> 1;
        off = (s & 1u) ? b_off[i] : a_off[i];
        len = (s & 1u) ? b_len[i] : a_len[i];
        const bool inside = off <= n_syms && (uint64_t)len <= n_syms - off;
        return inside ? (len + 31u) / 32u : 0xffffffffu;
    };
    uint32_t sum = 0;
    for (uint32_t s = s0; s < s1; s++) {
        uint64_t off; uint32_t len;
        const uint32_t nb = blocks_of(s, off, len);
        if (nb != 0xffffffffu) sum = min(sum + nb, 0x40000000u);         // saturates far above any capacity
    }
    part[t] = sum;
    __syncthreads();
    for (uint32_t d = 1; d < 1024; d <<= 1) {                            // inclusive scan of the 1024 partial sums
        const uint32_t v = t >= d ? part[t - d] : 0u;
        __syncthreads();
        part[t] = min(part[t] + v, 0x40000000u);
        __syncthreads();
    }
    uint32_t at = t ? part[t - 1] : 0u;
    for (uint32_t s = s0; s < s1; s++) {
        uint64_t off; uint32_t len;
        const uint32_t nb = blocks_of(s, off, len);
        const bool ok = nb != 0xffffffffu && at + nb <= cap_blocks;      // outside the symbol buffer, or the arena is full
        seq_byte_off[s] = off;
        seq_len[s] = ok ? len : 0xffffffffu;
        seq_blk_off[s] = min(at, cap_blocks);
        if (nb != 0xffffffffu) at = min(at + nb, 0x40000000u);
    }
    if (t == 1023) sc[AT_COUNTERS + 16] = min(part[1023], cap_blocks);
}

// grid-stride over the blocks of the arena; the number of blocks is only known on the device
template <int NP>
__global__ void __launch_bounds__(256)
k_na_pack(const uint8_t *__restrict__ syms, uint32_t n_seqs, uint32_t n_symbols, uint32_t *__restrict__ sc)
{
    const uint64_t *seq_byte_off = reinterpret_cast<const uint64_t *>(sc + AT_SEQ_BYTE_OFF);
    uint32_t *seq_len = sc + AT_SEQ_LEN;
    const uint32_t *seq_blk_off = sc + AT_SEQ_BLK_OFF;
    const uint32_t total = sc[AT_COUNTERS + 16];
    for (uint32_t blk = blockIdx.x * blockDim.x + threadIdx.x; blk < total; blk += gridDim.x * blockDim.x) {
        uint32_t lo = 0, hi = n_seqs;                                    // last s with seq_blk_off[s] <= blk
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (seq_blk_off[mid] <= blk) lo = mid; else hi = mid;
        }
        const uint32_t len = seq_len[lo];
        if (len == 0xffffffffu) continue;                                // a sequence that straddles the end of the arena
        const uint32_t k = blk - seq_blk_off[lo];
        uint32_t out[NP];
        const uint8_t *src = syms + seq_byte_off[lo] + (uint64_t)k * 32u;
        const uint32_t cnt = min(32u, len - k * 32u);
        const uint32_t seen = malcs_dev::pack_block<NP>(src, cnt, out);      // the OR of the block's codes: an upper bound of every one
#pragma unroll
        for (int p = 0; p < NP; p++) sc[AT_PLANES + (size_t)blk * NP + p] = out[p];
        if (seen >= n_symbols) {
            // a code that is not below n_symbols (malcs_batch_create rejects the batch; here the pairs of the sequence get
            // -1): beyond the bit planes it would alias another symbol and the distance would be silently wrong
            uint32_t worst = 0;
            for (uint32_t j = 0; j < cnt; j++) worst = max(worst, (uint32_t)src[j]);
            if (worst >= n_symbols) seq_len[lo] = 0xffffffffu;           // (only ever set, read again by k_na_classify)
        }
    }
}

// the launch class of malcs_dist_plan (malcs_dist.hip) for a pattern of m symbols, as an index into the lists
__device__ __forceinline__ uint32_t na_class(uint32_t m)
{
    const uint32_t words = max(1u, (m + 63u) / 64u);
    if (words <= 1) return 0;
    if (words <= 2) return 1;
    if (words <= 4) return 2;
    const uint32_t lanes = (words + 3u) / 4u;                            // W = 4: lanes rounded up to a power of two
    return 3u + (32u - __clz(lanes - 1u)) - 1u;                          // 2 -> 3, 3..4 -> 4, ..., 33..64 -> 8
}

__global__ void __launch_bounds__(256)
k_na_classify(uint32_t n_pairs, int NP, uint32_t *__restrict__ sc, int32_t *__restrict__ dist)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pairs) return;
    const uint32_t *seq_len = sc + AT_SEQ_LEN, *seq_blk_off = sc + AT_SEQ_BLK_OFF;
    const uint32_t la = seq_len[2 * i], lb = seq_len[2 * i + 1];
    const uint32_t m = min(la, lb);
    if (la == 0xffffffffu || lb == 0xffffffffu || m > MAXLEN) { dist[i] = -1; return; }
    uint32_t *pa = sc + AT_PAIR;
    pa[i] = seq_blk_off[2 * i] * (uint32_t)NP; pa[MAXP + i] = la;
    pa[2 * MAXP + i] = seq_blk_off[2 * i + 1] * (uint32_t)NP; pa[3 * MAXP + i] = lb;
    const uint32_t cls = na_class(m);
    const uint32_t idx = atomicAdd(&sc[AT_COUNTERS + cls], 1u);
    sc[AT_LISTS + (size_t)cls * MAXP + idx] = i;
}

}  // namespace

// Once per context, with its device current: the address of this device's copy of the scratch array.
int malcs_noalloc_setup(malcs_ctx *c)
{
    void *p = nullptr;
    const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
    if (e != hipSuccess || !p) {
        malcs_set_error("hipGetSymbolAddress(scratch) failed on device %d: %s", c->device, hipGetErrorString(e));
        return MALCS_ERR_HIP;
    }
    c->noalloc_scratch = static_cast<uint32_t *>(p);
    return MALCS_OK;
}

extern "C" int malcs_edit_distance_noalloc(malcs_ctx *ctx, void *stream, const uint8_t *d_syms, uint64_t n_syms,
                                           const uint64_t *d_a_off, const uint32_t *d_a_len,
                                           const uint64_t *d_b_off, const uint32_t *d_b_len,
                                           uint64_t n_pairs, int n_symbols, int32_t *d_dist)
{
    if (!ctx) { malcs_set_error("ctx is NULL"); return MALCS_ERR_ARG; }
    if (n_pairs > MAXP) { malcs_set_error("malcs_edit_distance_noalloc takes at most %u pairs per call, got %llu", MAXP, (unsigned long long)n_pairs); return MALCS_ERR_TOO_LONG; }
    if (n_symbols < 1 || n_symbols > 16) { malcs_set_error("malcs_edit_distance_noalloc takes 1 to 16 distinct symbols, got %d", n_symbols); return MALCS_ERR_ARG; }
    if (n_syms > MALCS_NOALLOC_MAX_SYMS) { malcs_set_error("malcs_edit_distance_noalloc takes a symbol buffer of at most %llu bytes, got %llu", (unsigned long long)MALCS_NOALLOC_MAX_SYMS, (unsigned long long)n_syms); return MALCS_ERR_TOO_LONG; }
    if (n_pairs == 0) return MALCS_OK;
    if (!d_a_off || !d_a_len || !d_b_off || !d_b_len || !d_dist || (n_syms && !d_syms)) { malcs_set_error("NULL argument"); return MALCS_ERR_ARG; }
    if (!ctx->noalloc_scratch) { malcs_set_error("the context has no scratch address"); return MALCS_ERR_INTERNAL; }
    MALCS_HIP(hipSetDevice(ctx->device));
    hipStream_t st = stream ? static_cast<hipStream_t>(stream) : ctx->stream;
    uint32_t *sc = ctx->noalloc_scratch;                                 // the device address, never the array's name
    const int NP = n_symbols <= 4 ? 2 : (n_symbols <= 8 ? 3 : 4);
    const uint32_t np32 = (uint32_t)n_pairs, cap_blocks = PLANE_DWORDS / (uint32_t)NP;
    hipLaunchKernelGGL(k_na_scan, dim3(1), dim3(1024), 0, st, d_a_off, d_a_len, d_b_off, d_b_len, np32, n_syms, cap_blocks, sc);
    const dim3 pack_grid(2048), b256(256);
    switch (NP) {
    case 2: hipLaunchKernelGGL(k_na_pack<2>, pack_grid, b256, 0, st, d_syms, 2u * np32, (uint32_t)n_symbols, sc); break;
    case 3: hipLaunchKernelGGL(k_na_pack<3>, pack_grid, b256, 0, st, d_syms, 2u * np32, (uint32_t)n_symbols, sc); break;
    case 4: hipLaunchKernelGGL(k_na_pack<4>, pack_grid, b256, 0, st, d_syms, 2u * np32, (uint32_t)n_symbols, sc); break;
    }
    hipLaunchKernelGGL(k_na_classify, dim3((np32 + 255u) / 256u), b256, 0, st, np32, NP, sc, d_dist);
    MALCS_HIP(hipGetLastError());
    const uint32_t *pa = sc + AT_PAIR;
    for (uint32_t c = 0; c < N_CLASSES; c++) {
        const int W = c == 0 ? 1 : (c == 1 ? 2 : 4), L = c < 3 ? 1 : 1 << (c - 2);
        const uint32_t ppw = 64u / (uint32_t)L;
        const int rc = malcs_dist_launch_list(st, NP, W, L, sc + AT_PLANES, pa, pa + MAXP, pa + 2 * MAXP, pa + 3 * MAXP,
                                              sc + AT_LISTS, c * MAXP, sc + AT_COUNTERS + c, MAXP,
                                              std::min<uint32_t>((np32 + ppw - 1u) / ppw, 1024u), d_dist);
        if (rc) return rc;
    }
    return MALCS_OK;
}
