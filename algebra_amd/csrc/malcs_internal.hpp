// malcs_internal.hpp -- shared host-side declarations of libmalcs (not part of the C-ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>
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

struct malcs_ctx {
    int device = -1;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr;
    hipDeviceProp_t prop;
};

// One launch of a family-1 kernel: a class of pairs that share (words per lane, lanes per pair).
struct dist_bucket {
    int W = 0;              // 64-bit words per lane
    int L = 0;              // lanes per pair (power of two)
    uint32_t first = 0;     // range [first, first + count) of batch->order
    uint32_t count = 0;
};

struct malcs_batch {
    malcs_ctx *ctx = nullptr;
    uint64_t n_pairs = 0;
    int n_planes = 2;
    // device
    uint32_t *d_planes = nullptr;        // packed arena: per sequence, per 32-symbol block, n_planes dwords
    uint32_t *d_a_off = nullptr, *d_a_len = nullptr, *d_b_off = nullptr, *d_b_len = nullptr;   // per pair (dword offsets)
    uint32_t *d_order = nullptr;         // pair ids grouped by bucket
    int32_t *d_dist = nullptr;
    // host mirrors
    std::vector<uint32_t> a_off, a_len, b_off, b_len;   // dword offsets into d_planes / lengths in symbols
    std::vector<dist_bucket> dist_buckets;
    std::vector<uint32_t> band_pairs;    // pairs too long for a resident pattern: sliding-window kernel
    uint64_t arena_dwords = 0;
    uint32_t max_len = 0;
};

// malcs_pack.hip
int malcs_pack_launch(malcs_ctx *ctx, const uint8_t *d_syms, const uint64_t *d_seq_byte_off,
                      const uint32_t *d_seq_len, const uint32_t *d_seq_blk_off, uint32_t n_seqs,
                      uint32_t total_blocks, int n_planes, uint32_t *d_planes);

// malcs_dist.hip
int malcs_dist_plan(malcs_batch *b);
int malcs_dist_launch(malcs_batch *b);
int malcs_dist_band(malcs_batch *b, const std::vector<uint32_t> &pairs, int32_t *host_out);
