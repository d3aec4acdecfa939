// malcs_dist.hip -- kernel family 1: batched bit-parallel LLCS -> indel-only edit distance.
//
// Stands in for edit_distance(a, b) (reference algebra/lcs/edit_distance.py:21-53), which returns
// m + n - 2 * LLCS(a, b).  The reference walks an O(NP) furthest-point sweep; here every pair runs
// the bit-vector LCS recurrence over the full matrix, 32 cells per VALU instruction:
//
//     U  = V & Match[text symbol]          Match built on the fly from the pattern's bit planes
//     V' = (V + U) | (V ^ U)               (V ^ U == V & ~Match because U is a subset of V)
//     LLCS = number of zero bits of V among the first m positions after the last text symbol
//
// Mapping to the machine (gfx950, wave64):
//   * the shorter sequence is the pattern (rows, bit positions), the longer one the text (columns);
//   * a pair owns L = 2^k adjacent lanes of a wave, each lane owns W 64-bit words (2W dword limbs) of
//     the pattern in VGPRs: bit planes P[plane][limb] and the state V[limb]; 64/L pairs share a wave;
//   * one column costs, per limb, NP v_bitop3_b32 (U), one v_addc_co_u32 (the sum, carry chained
//     through the lane's limbs) and one v_bitop3_b32 (V');
//   * the lanes of a pair form a systolic array: lane q works on text column (step - q), so the carry
//     out of lane q-1 is consumed by lane q one step later.  Carries live in SGPR pairs (lane masks):
//     limb to limb through v_addc_co_u32, lane to lane as (carry_out << 1) & segmask on the scalar
//     unit.  No look-ahead, no second pass, exact for any carry length (carries routinely run through
//     every lane above the DP frontier);
//   * text symbols come from the packed arena, NP dwords per 32 columns and lane.
//
// Algorithmic work: 4 dword ops per 32 cells for DNA = 0.125 lane-op per cell.
#include "malcs_internal.hpp"
#include "malcs_device.hpp"
#include "malcs_bitpar.hpp"
#include "malcs_hostpack.hpp"
#include <algorithm>

namespace {

using namespace malcs_dev;

template <int W, int NP, bool XL>
__global__ void __launch_bounds__(256, MALCS_MINW)
k_llcs_bitpar(const uint32_t *__restrict__ planes,
              const uint32_t *__restrict__ a_off, const uint32_t *__restrict__ a_len,
              const uint32_t *__restrict__ b_off, const uint32_t *__restrict__ b_len,
              const uint32_t *__restrict__ order, uint32_t first, uint32_t count, int logL,
              int32_t *__restrict__ dist)
{
    llcs_bitpar_wave<W, NP, XL>(planes, a_off, a_len, b_off, b_len, order, first, count, logL, dist,
                                (blockIdx.x * blockDim.x + threadIdx.x) >> 6);
}

// The same waves over a work list whose length is only known on the device (the pairs the window
// kernel of malcs_diag.hip could not settle): a fixed grid, every wave takes slots until the list is
// exhausted.  The count is read once, so every wave leaves the loop.
template <int W, int NP, bool XL>
__global__ void __launch_bounds__(256, 6)          // the loop around the wave costs a few registers more
k_llcs_bitpar_list(const uint32_t *__restrict__ planes,
                   const uint32_t *__restrict__ a_off, const uint32_t *__restrict__ a_len,
                   const uint32_t *__restrict__ b_off, const uint32_t *__restrict__ b_len,
                   const uint32_t *__restrict__ list, uint32_t first, const uint32_t *__restrict__ count_ptr,
                   uint32_t cap, int logL, int32_t *__restrict__ dist)
{
    uint32_t count = __builtin_amdgcn_readfirstlane(*count_ptr);
    if (count > cap) count = cap;
    const unsigned ppw = 64u >> logL;
    const unsigned nwaves = (gridDim.x * blockDim.x) >> 6;
    for (unsigned wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; wave * ppw < count; wave += nwaves)
        llcs_bitpar_wave<W, NP, XL>(planes, a_off, a_len, b_off, b_len, list, first, count, logL, dist, wave);
}

// ------------------------------------------------------------------------------------------------
// Banded variant for long, similar pairs (shorter sequence beyond what 64 lanes x 16 words hold, or
// simply long): one wave per pair, the 64 lanes hold a window of 64 row groups (G = 64*W rows each)
// that slides down the matrix with the band of diagonals an alignment of cost <= T can visit
// (k = col - row in [-(T-delta)/2, delta + (T-delta)/2], delta = n - m >= 0).
//
// Row group u always works on text column (step - u) -- the same systolic schedule as above, so the
// carry chain is unchanged.  Every 32 steps the window may advance by one group: registers move one
// lane down, the group that leaves is frozen (its zero bits are banked), a fresh group (V = all ones)
// enters in lane 63.  Freezing a group and starting a group late both restrict the DP to paths that
// stay inside the window, so the result D' is exact whenever D' <= T (then no alignment of cost <= D'
// leaves the band); the host retries with a wider window otherwise.
template <int W, int NP>
__global__ void __launch_bounds__(64)
k_llcs_band(const uint32_t *__restrict__ planes,
            const uint32_t *__restrict__ a_off, const uint32_t *__restrict__ a_len,
            const uint32_t *__restrict__ b_off, const uint32_t *__restrict__ b_len,
            const uint32_t *__restrict__ work, const int32_t *__restrict__ thresh, int32_t *__restrict__ out)
{
    constexpr int NL = 2 * W;
    constexpr int G = 64 * W;
    constexpr int UC = NL <= 4 ? 16 : (NL <= 8 ? 8 : 4);
    const unsigned lane = threadIdx.x;
    const uint32_t pid = work[blockIdx.x];
    const uint32_t la = a_len[pid], lb = b_len[pid];
    uint32_t m, n, po, to;
    if (la <= lb) { m = la; n = lb; po = a_off[pid]; to = b_off[pid]; }
    else          { m = lb; n = la; po = b_off[pid]; to = a_off[pid]; }
    const int T = thresh[blockIdx.x];
    const int delta = (int)(n - m);
    const int kmax = delta + (T > delta ? (T - delta) / 2 : 0);
    const uint32_t pblk = (m + 31u) >> 5;
    const int tblk = (int)((n + 31u) >> 5);

    // lowest row group still needed at step t0
    auto sigma_at = [&](int t0) -> int { const int v = t0 - G + 1 - kmax; return v <= 0 ? 0 : (v + G) / (G + 1); };
    auto load_group = [&](uint32_t u, uint32_t (&P)[NL][NP]) {
#pragma unroll
        for (int k = 0; k < NL; k++) {
            const uint32_t g = u * NL + k;
            const bool in = g < pblk;
#pragma unroll
            for (int p = 0; p < NP; p++) P[k][p] = in ? planes[po + g * NP + p] : 0u;
        }
    };
    auto zeros_of = [&](uint32_t u, const uint32_t (&V)[NL]) -> uint32_t {
        uint32_t z = 0;
#pragma unroll
        for (int k = 0; k < NL; k++) {
            const long long vb = (long long)m - (long long)(u * NL + k) * 32ll;
            const uint32_t mask = vb >= 32 ? ~0u : (vb <= 0 ? 0u : ((1u << vb) - 1u));
            z += __popc(~V[k] & mask);
        }
        return z;
    };
    auto load_text = [&](int B, uint32_t u, uint32_t (&tc)[NP], uint32_t (&tp)[NP]) {
        const int blk = B - (int)(u >> 5);
#pragma unroll
        for (int p = 0; p < NP; p++) {
            tc[p] = (blk >= 0 && blk < tblk) ? planes[to + (uint32_t)blk * NP + p] : 0u;
            tp[p] = (blk >= 1 && blk - 1 < tblk) ? planes[to + (uint32_t)(blk - 1) * NP + p] : 0u;
        }
    };

    int sigma = 0;
    uint32_t u = lane;
    uint32_t P[NL][NP], V[NL];
    load_group(u, P);
#pragma unroll
    for (int k = 0; k < NL; k++) V[k] = ~0u;
    uint32_t frozen = 0;
    const uint64_t segmask = ~1ull;
    uint64_t cin;
    asm("s_mov_b64 %0, 0" : "=s"(cin));

    const int u_last = m ? (int)((m - 1u) / G) : 0;
    const int t_end = (int)n - 1 + u_last;
    uint32_t tc[NP], tp[NP];
    load_text(0, u, tc, tp);

    for (int B = 0; 32 * B <= t_end; B++) {
        const int sigma_next = sigma_at(32 * (B + 1));
        uint32_t tcn[NP], tpn[NP];
        load_text(B + 1, (uint32_t)sigma_next + lane, tcn, tpn);      // prefetch for the next block
        const uint32_t qq = u & 31u;
        uint32_t w[NP];
#pragma unroll
        for (int p = 0; p < NP; p++) w[p] = qq ? __builtin_amdgcn_alignbit(tc[p], tp[p], 32u - qq) : tc[p];
        const long long lo = (long long)u - 32ll * B;
        const long long hi = (long long)n + (long long)u - 32ll * B;
        const uint32_t mlo = lo <= 0 ? ~0u : (lo >= 32 ? 0u : (~0u << lo));
        const uint32_t mhi = hi >= 32 ? ~0u : (hi <= 0 ? 0u : ((1u << hi) - 1u));
        const uint32_t vm = mlo & mhi;
        if (__builtin_amdgcn_ballot_w64(vm != ~0u) == 0) {
#pragma unroll UC
            for (int s = 0; s < 32; s++) {
                uint32_t x[NP];
#pragma unroll
                for (int p = 0; p < NP; p++) x[p] = (uint32_t)__builtin_amdgcn_sbfe((int)w[p], s, 1);
                const uint64_t co = column_step<NL, NP, false>(V, P, x, 0u, cin);
                cin = carry_to_next_lane(co, segmask);
            }
        } else {
#pragma unroll UC
            for (int s = 0; s < 32; s++) {
                uint32_t x[NP];
#pragma unroll
                for (int p = 0; p < NP; p++) x[p] = (uint32_t)__builtin_amdgcn_sbfe((int)w[p], s, 1);
                const uint32_t act = (uint32_t)__builtin_amdgcn_sbfe((int)vm, s, 1);
                const uint64_t co = column_step<NL, NP, true>(V, P, x, act, cin);
                cin = carry_to_next_lane(co, segmask);
            }
        }
        // slide the window: the top group is frozen, everything moves one lane down
        for (; sigma < sigma_next; sigma++) {
            if (lane == 0) frozen += zeros_of(u, V);
#pragma unroll
            for (int k = 0; k < NL; k++) {
                V[k] = __shfl_down(V[k], 1);
#pragma unroll
                for (int p = 0; p < NP; p++) P[k][p] = __shfl_down(P[k][p], 1);
            }
            u += 1;
            if (lane == 63) {
                load_group(u, P);
#pragma unroll
                for (int k = 0; k < NL; k++) V[k] = ~0u;
            }
            cin = carry_window_down(cin);
        }
#pragma unroll
        for (int p = 0; p < NP; p++) { tc[p] = tcn[p]; tp[p] = tpn[p]; }
    }

    uint32_t zeros = frozen + zeros_of(u, V);
    for (unsigned s = 1; s < 64; s <<= 1) zeros += __shfl_xor(zeros, s);
    if (lane == 0) out[blockIdx.x] = (int32_t)(m + n - 2u * zeros);
}

typedef void (*band_kernel_t)(const uint32_t *, const uint32_t *, const uint32_t *, const uint32_t *, const uint32_t *,
                              const uint32_t *, const int32_t *, int32_t *);
template <int NP>
band_kernel_t pick_band(int W)
{
    switch (W) {
    case 1: return k_llcs_band<1, NP>;
    case 2: return k_llcs_band<2, NP>;
    case 4: return k_llcs_band<4, NP>;
    case 8: if constexpr (NP <= 4) return k_llcs_band<8, NP>; else return nullptr;
    case 16: if constexpr (NP <= 3) return k_llcs_band<16, NP>; else return nullptr;
    default: return nullptr;
    }
}

typedef void (*dist_kernel_t)(const uint32_t *, const uint32_t *, const uint32_t *, const uint32_t *,
                              const uint32_t *, const uint32_t *, uint32_t, uint32_t, int, int32_t *);
typedef void (*list_kernel_t)(const uint32_t *, const uint32_t *, const uint32_t *, const uint32_t *,
                              const uint32_t *, const uint32_t *, uint32_t, const uint32_t *, uint32_t, int, int32_t *);

template <int NP>
list_kernel_t pick_list_kernel(int W, bool xl)
{
    switch (W) {
    case 1: return xl ? k_llcs_bitpar_list<1, NP, true> : k_llcs_bitpar_list<1, NP, false>;
    case 2: return xl ? k_llcs_bitpar_list<2, NP, true> : k_llcs_bitpar_list<2, NP, false>;
    case 4: return xl ? k_llcs_bitpar_list<4, NP, true> : k_llcs_bitpar_list<4, NP, false>;
    case 8: if constexpr (NP <= 4) return k_llcs_bitpar_list<8, NP, true>; else return nullptr;
    case 16: if constexpr (NP <= 3) return k_llcs_bitpar_list<16, NP, true>; else return nullptr;
    default: return nullptr;
    }
}

template <int NP>
dist_kernel_t pick_kernel(int W, bool xl)
{
    switch (W) {
    case 1: return xl ? k_llcs_bitpar<1, NP, true> : k_llcs_bitpar<1, NP, false>;
    case 2: return xl ? k_llcs_bitpar<2, NP, true> : k_llcs_bitpar<2, NP, false>;
    case 4: return xl ? k_llcs_bitpar<4, NP, true> : k_llcs_bitpar<4, NP, false>;
    case 8: if constexpr (NP <= 4) return k_llcs_bitpar<8, NP, true>; else return nullptr;
    case 16: if constexpr (NP <= 3) return k_llcs_bitpar<16, NP, true>; else return nullptr;
    default: return nullptr;
    }
}

int max_words_per_lane(int n_planes) { return n_planes <= 3 ? 16 : (n_planes == 4 ? 8 : 4); }

uint32_t pow2ceil(uint32_t x) { uint32_t p = 1; while (p < x) p <<= 1; return p; }

}  // namespace

// One full-matrix launch over a work list whose length sits in device memory (*d_count, read by the kernel):
// pairs list[first .. first + min(*d_count, cap)) in class (W words per lane, L lanes per pair), `waves` waves.
int malcs_dist_launch_list(hipStream_t st, int n_planes, int W, int L, const uint32_t *d_planes,
                           const uint32_t *d_a_off, const uint32_t *d_a_len, const uint32_t *d_b_off, const uint32_t *d_b_len,
                           const uint32_t *d_list, uint32_t first, const uint32_t *d_count, uint32_t cap, uint32_t waves,
                           int32_t *d_dist)
{
    int logL = 0;
    while ((1 << logL) < L) logL++;
    list_kernel_t fn = nullptr;
    switch (n_planes) {
    case 2: fn = pick_list_kernel<2>(W, L > 1); break;
    case 3: fn = pick_list_kernel<3>(W, L > 1); break;
    case 4: fn = pick_list_kernel<4>(W, L > 1); break;
    }
    if (!fn) { malcs_set_error("no distance kernel for W=%d planes=%d", W, n_planes); return MALCS_ERR_INTERNAL; }
    hipLaunchKernelGGL(fn, dim3((std::max(waves, 1u) + 3u) / 4u), dim3(256), 0, st, d_planes, d_a_off, d_a_len,
                       d_b_off, d_b_len, d_list, first, d_count, cap, logL, d_dist);
    MALCS_HIP(hipGetLastError());
    return MALCS_OK;
}

// Group the pairs of a batch into launch classes (W words per lane, L lanes per pair).
int malcs_dist_plan(malcs_batch *b)
{
    const uint64_t n = b->n_pairs;
    malcs_stage_trace trace("distance plan");
    b->band_pairs.clear();
    const int wmax = max_words_per_lane(b->n_planes);
    // class id = index into table of (W, logL); 255 = beyond a resident pattern, 254 = window class
    struct wl { int W, logL; };
    std::vector<wl> table;
    auto class_of = [&](int W, int logL) -> uint8_t {
        for (size_t i = 0; i < table.size(); i++) if (table[i].W == W && table[i].logL == logL) return (uint8_t)i;
        table.push_back({W, logL});
        return (uint8_t)(table.size() - 1);
    };
    // Window class (malcs_diag.hip): pairs long enough that a few limbs around the diagonal are much
    // less work than the whole matrix, and whose length difference alone does not already exceed what
    // the window can prove.  Whatever it cannot settle returns to the class chosen here.
    int nlw = 3;
    if (const char *e = getenv("MALCS_DIAG_NLW")) nlw = atoi(e);
    const bool window = nlw > 0 && b->n_planes <= 4;
    const uint32_t T = window ? (uint32_t)malcs_diag_threshold(nlw) : 0u;
    std::vector<uint8_t> &cls = b->plan_cls, &in_window = b->plan_in_window;
    cls.resize(n); in_window.resize(n);
    uint64_t cells = 0, n_window = 0;
    // the class of a pair follows from the 64-bit words of its shorter side alone
    auto class_of_words = [&](uint32_t words) -> uint8_t {
        int W, logL = 0;
        if (words <= 4 && (int)words <= wmax) return class_of((int)pow2ceil(words), 0);
        W = 4;
        while ((uint32_t)W * 64u < words && W < wmax) W *= 2;
        if ((uint32_t)W * 64u < words) return 255;                  // too long for a resident pattern: sliding-window kernel
        const uint32_t lanes = pow2ceil((words + (uint32_t)W - 1u) / (uint32_t)W);
        while ((1u << logL) < lanes) logL++;
        return class_of(W, logL);
    };
    auto classify = [&](uint64_t lo, uint64_t hi, const uint8_t *by_words, uint64_t &cells_out, uint64_t &window_out, std::vector<uint32_t> &band) {
        uint64_t cl = 0, nw = 0;
        uint32_t longest = 0;                                       // of the window class (sizes its plan's histogram)
        uint64_t blocks = 0;                                        // ... and the 32-column blocks of its texts (size its row buffer)
        uint32_t last_words = 0xffffffffu; uint8_t last_cls = 0;    // batches of equal lengths: one lookup
        for (uint64_t i = lo; i < hi; i++) {
            const uint32_t la = b->a_len[i], lb = b->b_len[i];
            const uint32_t m = std::min(la, lb), mx = std::max(la, lb);
            cl += (uint64_t)la * lb;
            const uint32_t words = std::max<uint32_t>(1u, (m + 63u) / 64u);
            if (words != last_words) { last_words = words; last_cls = by_words ? by_words[words] : class_of_words(words); }
            cls[i] = last_cls;
            in_window[i] = 0;
            if (last_cls == 255) { band.push_back((uint32_t)i); continue; }
            if (window && (m + 31u) / 32u >= 2u * (uint32_t)nlw + 2u && mx - m <= T) { in_window[i] = 1; nw++; longest = std::max(longest, mx); blocks += (mx + 31u) / 32u; }
        }
        cells_out = cl; window_out = nw;
        uint32_t seen = __atomic_load_n(&b->plan_window_longest, __ATOMIC_RELAXED);
        while (longest > seen && !__atomic_compare_exchange_n(&b->plan_window_longest, &seen, longest, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
        __atomic_fetch_add(&b->plan_window_blocks, blocks, __ATOMIC_RELAXED);
    };
    b->plan_window_longest = 0;
    b->plan_window_blocks = 0;
    malcs_workers *wk = n >= 16384 ? malcs_ctx_workers(b->ctx) : nullptr;
    const int NT = wk ? wk->size() : 1;
    std::vector<uint64_t> part_window((size_t)NT, 0);
    auto range = [&](int k, uint64_t &lo, uint64_t &hi) { lo = n * (uint64_t)k / (uint64_t)NT; hi = n * (uint64_t)(k + 1) / (uint64_t)NT; };
    if (wk) {
        // large batches, on the context's threads: the classes of all word counts up to the longest sequence first (the
        // table is shared), then the pairs in stretches
        const uint32_t max_words = std::max<uint32_t>(1u, (b->max_len + 63u) / 64u);
        std::vector<uint8_t> seen((size_t)max_words + 1, 0), by_words((size_t)max_words + 1, 0);
        wk->run(NT, [&](int k) {
            uint64_t lo, hi; range(k, lo, hi);
            for (uint64_t i = lo; i < hi; i++)
                __atomic_store_n(&seen[std::max<uint32_t>(1u, (std::min(b->a_len[i], b->b_len[i]) + 63u) / 64u)], (uint8_t)1, __ATOMIC_RELAXED);
        });
        for (uint32_t w = 1; w <= max_words; w++) if (seen[w]) by_words[w] = class_of_words(w);
        std::vector<uint64_t> part_cells((size_t)NT, 0);
        std::vector<std::vector<uint32_t>> part_band((size_t)NT);
        wk->run(NT, [&](int k) {
            uint64_t lo, hi; range(k, lo, hi);
            classify(lo, hi, by_words.data(), part_cells[(size_t)k], part_window[(size_t)k], part_band[(size_t)k]);
        });
        for (int k = 0; k < NT; k++) {
            cells += part_cells[(size_t)k]; n_window += part_window[(size_t)k];
            b->band_pairs.insert(b->band_pairs.end(), part_band[(size_t)k].begin(), part_band[(size_t)k].end());
        }
    } else {
        classify(0, n, nullptr, cells, n_window, b->band_pairs);
        part_window[0] = n_window;
    }
    b->full_cells = cells;
    malcs_diag_release(b);
    if (n_window >= 64) {
        std::vector<uint32_t> &diag_pairs = b->plan_diag_pairs;
        std::vector<uint8_t> &diag_cls = b->plan_diag_cls;
        diag_pairs.resize(n_window); diag_cls.resize(n_window);
        std::vector<int> retry_id(table.size(), -1);
        std::vector<diag_retry> classes;
        if (wk) {
            // which classes have pairs in the window class (ids in the order of the table), then the lists in stretches
            std::vector<uint8_t> used(table.size() * (size_t)NT, 0);
            wk->run(NT, [&](int k) {
                uint64_t lo, hi; range(k, lo, hi);
                uint8_t *u = used.data() + table.size() * (size_t)k;
                for (uint64_t i = lo; i < hi; i++) if (in_window[i]) u[cls[i]] = 1;
            });
            for (size_t c = 0; c < table.size(); c++) {
                bool any = false;
                for (int k = 0; k < NT; k++) any = any || used[table.size() * (size_t)k + c];
                if (any) { retry_id[c] = (int)classes.size(); diag_retry r; r.W = table[c].W; r.L = 1 << table[c].logL; classes.push_back(r); }
            }
            std::vector<uint64_t> first((size_t)NT + 1, 0);
            for (int k = 0; k < NT; k++) first[(size_t)k + 1] = first[(size_t)k] + part_window[(size_t)k];
            wk->run(NT, [&](int k) {
                uint64_t lo, hi; range(k, lo, hi);
                size_t at = first[(size_t)k];
                for (uint64_t i = lo; i < hi; i++) {
                    if (!in_window[i]) continue;
                    diag_pairs[at] = (uint32_t)i; diag_cls[at] = (uint8_t)retry_id[cls[i]]; at++;
                    cls[i] = 254;                          // not in the direct launches
                }
            });
        } else {
            size_t k = 0;
            for (uint64_t i = 0; i < n; i++) {
                if (!in_window[i]) continue;
                int &id = retry_id[cls[i]];
                if (id < 0) { id = (int)classes.size(); diag_retry r; r.W = table[cls[i]].W; r.L = 1 << table[cls[i]].logL; classes.push_back(r); }
                diag_pairs[k] = (uint32_t)i; diag_cls[k] = (uint8_t)id; k++;
                cls[i] = 254;                              // not in the direct launches
            }
        }
        trace.lap("classes");
        int rc = malcs_diag_setup(b, nlw, diag_pairs, diag_cls, classes);
        if (!rc) rc = malcs_diag_plan(b);
        if (rc) return rc;
    }

    std::vector<uint32_t> cnt(table.size() + 1, 0);
    for (uint64_t i = 0; i < n; i++) if (cls[i] < 254) cnt[cls[i] + 1]++;
    for (size_t c = 0; c < table.size(); c++) cnt[c + 1] += cnt[c];
    const uint32_t n_direct = cnt[table.size()];
    b->dist_buckets.clear();
    for (size_t c = 0; c < table.size(); c++) {
        dist_bucket k;
        k.W = table[c].W; k.L = 1 << table[c].logL; k.first = cnt[c]; k.count = cnt[c + 1] - cnt[c];
        if (k.count) b->dist_buckets.push_back(k);
    }
    if (n_direct) {
        std::vector<uint32_t> order(n_direct), pos(cnt.begin(), cnt.end() - 1);
        for (uint64_t i = 0; i < n; i++) if (cls[i] < 254) order[pos[cls[i]]++] = (uint32_t)i;
        MALCS_HIP(malcs_h2d(b->ctx->stream, b->d_order, order.data(), (size_t)n_direct * sizeof(uint32_t)));
        MALCS_HIP(malcs_sync(b->ctx->stream));       // `order` is a local
    } else if (n) {
        MALCS_HIP(malcs_sync(b->ctx->stream));       // batch creation ends with the stream drained (the context's pinned buffers are idle again)
    }
    trace.lap("direct classes");
    return MALCS_OK;
}

// What depends on the sequences themselves (the classes above follow from the lengths): the plan of the window class,
// queued on the context's stream -- nothing comes back to the host.
int malcs_dist_replan(malcs_batch *b) { return malcs_diag_plan(b); }

int malcs_dist_launch(malcs_batch *b, hipEvent_t window_start, hipEvent_t window_stop)
{
    int rc = malcs_dist_launch_resident(b, window_start, window_stop);
    if (rc) return rc;
    if (!b->band_pairs.empty()) return malcs_dist_band(b, b->band_pairs, nullptr);
    return MALCS_OK;
}

// everything but the pairs beyond a resident pattern (b->band_pairs)
int malcs_dist_launch_resident(malcs_batch *b, hipEvent_t window_start, hipEvent_t window_stop)
{
    // window class first, then the full-matrix kernel over whatever it left undecided
    if (b->diag.nlw) {
        int rc = malcs_diag_launch(b, true, window_start, window_stop);
        if (rc) return rc;
        for (size_t c = 0; c < b->diag.retry.size(); c++) {
            const diag_retry &r = b->diag.retry[c];
            if (!r.cap) continue;
            const uint32_t ppw = 64u / (uint32_t)r.L;
            const int rc2 = malcs_dist_launch_list(b->ctx->stream, b->n_planes, r.W, r.L, b->d_planes, b->d_a_off, b->d_a_len,
                                                   b->d_b_off, b->d_b_len, b->diag.d_retry_list, r.first,
                                                   b->diag.d_retry_count + b->diag.parity * std::max<size_t>(b->diag.retry.size(), 1) + c,
                                                   r.cap, std::min<uint32_t>((r.cap + ppw - 1u) / ppw, 1024u), b->d_dist);
            if (rc2) return rc2;
        }
    }
    for (const dist_bucket &k : b->dist_buckets) {
        int logL = 0;
        while ((1 << logL) < k.L) logL++;
        const bool xl = k.L > 1;
        dist_kernel_t fn = nullptr;
        switch (b->n_planes) {
        case 2: fn = pick_kernel<2>(k.W, xl); break;
        case 3: fn = pick_kernel<3>(k.W, xl); break;
        case 4: fn = pick_kernel<4>(k.W, xl); break;
        case 8: fn = pick_kernel<8>(k.W, xl); break;
        }
        if (!fn) { malcs_set_error("no distance kernel for W=%d planes=%d", k.W, b->n_planes); return MALCS_ERR_INTERNAL; }
        const uint32_t ppw = 64u / (uint32_t)k.L;
        const uint32_t waves = (k.count + ppw - 1u) / ppw;
        const uint32_t blocks = (waves + 3u) / 4u;
        hipLaunchKernelGGL(fn, dim3(blocks), dim3(256), 0, b->ctx->stream, b->d_planes, b->d_a_off, b->d_a_len,
                           b->d_b_off, b->d_b_len, b->d_order, k.first, k.count, logL, b->d_dist);
        MALCS_HIP(hipGetLastError());
    }
    return MALCS_OK;
}

// Largest threshold a window of 64 groups of G = 64*W rows covers (see k_llcs_band): the groups needed
// during one block of 32 steps must fit, u_max(t0 + 31) - u_min(t0) + 1 <= 64.
static int band_capacity(int W) { const int G = 64 * W; return 61 * (G + 1) - G - 64; }

// Distances of long pairs with the sliding-window kernel; widens the window until the result is exact.
// Synchronous.  Results go to b->d_dist[pid] and, if `host_out` is given, to host_out[k] for pairs[k].
int malcs_dist_band(malcs_batch *b, const std::vector<uint32_t> &pairs, int32_t *host_out)
{
    hipStream_t st = b->ctx->stream;
    const int wmax = max_words_per_lane(b->n_planes);
    std::vector<uint32_t> todo;
    std::vector<size_t> slot_of;
    std::vector<int32_t> result(pairs.size(), -1);
    {
        // similar pairs first: the furthest-reaching sweep (malcs_sweep.hip) costs O(n / 32 + D^2), not columns x band
        int32_t cap = 32768;
        if (const char *e = getenv("MALCS_SWEEP_DCAP")) cap = atoi(e);
        if (cap > 0) {
            int rc0 = malcs_sweep_distances(b, pairs, cap, result);
            if (rc0) return rc0;
        }
        for (size_t i = 0; i < pairs.size(); i++) if (result[i] < 0) { todo.push_back(pairs[i]); slot_of.push_back(i); }
    }
    dev_scoped<uint32_t> work_dev; dev_scoped<int32_t> thresh_dev, out_dev;
    uint32_t *&d_work = work_dev.p; int32_t *&d_thresh = thresh_dev.p, *&d_out = out_dev.p;
    MALCS_HIP(hipMalloc(&d_work, pairs.size() * 4)); MALCS_HIP(hipMalloc(&d_thresh, pairs.size() * 4)); MALCS_HIP(hipMalloc(&d_out, pairs.size() * 4));
    int rc = MALCS_OK;
    for (int W = 1; W <= wmax && !todo.empty(); W *= 2) {
        const int T = band_capacity(W);
        std::vector<uint32_t> work; std::vector<size_t> slots; std::vector<int32_t> thresh;
        std::vector<uint32_t> later; std::vector<size_t> later_slots;
        for (size_t i = 0; i < todo.size(); i++) {
            const uint32_t pid = todo[i];
            const int64_t delta = std::llabs((int64_t)b->a_len[pid] - (int64_t)b->b_len[pid]);
            if (delta > T) { later.push_back(pid); later_slots.push_back(slot_of[i]); continue; }
            work.push_back(pid); slots.push_back(slot_of[i]); thresh.push_back(T);
        }
        if (!work.empty()) {
            band_kernel_t fn = nullptr;
            switch (b->n_planes) {
            case 2: fn = pick_band<2>(W); break;
            case 3: fn = pick_band<3>(W); break;
            case 4: fn = pick_band<4>(W); break;
            case 8: fn = pick_band<8>(W); break;
            }
            if (!fn) { rc = MALCS_ERR_INTERNAL; malcs_set_error("no band kernel for W=%d planes=%d", W, b->n_planes); break; }
            std::vector<int32_t> got(work.size());
            malcs_h2d(st, d_work, work.data(), work.size() * 4);
            malcs_h2d(st, d_thresh, thresh.data(), work.size() * 4);
            hipLaunchKernelGGL(fn, dim3((uint32_t)work.size()), dim3(64), 0, st, b->d_planes, b->d_a_off, b->d_a_len,
                               b->d_b_off, b->d_b_len, d_work, d_thresh, d_out);
            malcs_d2h(st, got.data(), d_out, work.size() * 4);
            hipError_t e = malcs_sync(st);
            if (e != hipSuccess) { rc = MALCS_ERR_HIP; malcs_set_error("band kernel failed: %s", hipGetErrorString(e)); break; }
            for (size_t i = 0; i < work.size(); i++) {
                if (got[i] <= T) result[slots[i]] = got[i];               // exact: no cheaper alignment can leave the band
                else { later.push_back(work[i]); later_slots.push_back(slots[i]); }
            }
        }
        todo.swap(later); slot_of.swap(later_slots);
    }
    if (rc == MALCS_OK && !todo.empty()) {
        malcs_set_error("pair %u: sequences of %u and %u symbols differ by more than the widest window (%d edits with %d bit planes) covers",
                        todo[0], b->a_len[todo[0]], b->b_len[todo[0]], band_capacity(wmax), b->n_planes);
        rc = MALCS_ERR_TOO_LONG;
    }
    if (rc == MALCS_OK) {
        for (size_t i = 0; i < pairs.size(); i++) {
            malcs_h2d(st, b->d_dist + pairs[i], &result[i], 4);
            if (host_out) host_out[i] = result[i];
        }
        malcs_sync(st);
    }
    return rc;
}
