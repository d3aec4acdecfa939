// malcs_pack.hip -- device-side packing of uint8 symbol codes into the bit-plane arena.
//
// Arena layout (the "packed sequence arena" all kernels read): for every sequence, for every
// block of 32 symbols, n_planes consecutive dwords; bit j of plane p of block k is bit p of the
// code of symbol 32*k + j.  Symbols past the end of a sequence pack as code 0.
//
// One thread packs one block: 32 contiguous bytes in (two 16-byte loads when aligned, byte
// loads otherwise), n_planes dwords out.  Threads of a wave touch consecutive blocks, so both
// sides are coalesced.
#include "malcs_internal.hpp"
#include "malcs_device.hpp"

template <int NP>
__global__ void __launch_bounds__(256)
k_pack_planes(const uint8_t *__restrict__ syms, const uint64_t *__restrict__ seq_byte_off,
              const uint32_t *__restrict__ seq_len, const uint32_t *__restrict__ seq_blk_off,
              uint32_t n_seqs, uint32_t total_blocks, uint32_t *__restrict__ planes)
{
    uint32_t blk = blockIdx.x * blockDim.x + threadIdx.x;
    if (blk >= total_blocks) return;
    // sequence owning this block: last s with seq_blk_off[s] <= blk
    uint32_t lo = 0, hi = n_seqs;
    while (hi - lo > 1) {
        uint32_t mid = (lo + hi) >> 1;
        if (seq_blk_off[mid] <= blk) lo = mid; else hi = mid;
    }
    uint32_t k = blk - seq_blk_off[lo];
    uint32_t len = seq_len[lo];
    const uint8_t *src = syms + seq_byte_off[lo] + (uint64_t)k * 32u;
    uint32_t cnt = len - k * 32u;
    if (cnt > 32u) cnt = 32u;
    uint32_t out[NP];
    malcs_dev::pack_block<NP>(src, cnt, out);
#pragma unroll
    for (int p = 0; p < NP; p++) planes[(uint64_t)blk * NP + p] = out[p];
}

int malcs_pack_launch(malcs_ctx *ctx, const uint8_t *d_syms, const uint64_t *d_seq_byte_off,
                      const uint32_t *d_seq_len, const uint32_t *d_seq_blk_off, uint32_t n_seqs,
                      uint32_t total_blocks, int n_planes, uint32_t *d_planes)
{
    if (total_blocks == 0) return MALCS_OK;
    dim3 grid((total_blocks + 255u) / 256u), block(256);
    switch (n_planes) {
    case 2: k_pack_planes<2><<<grid, block, 0, ctx->stream>>>(d_syms, d_seq_byte_off, d_seq_len, d_seq_blk_off, n_seqs, total_blocks, d_planes); break;
    case 3: k_pack_planes<3><<<grid, block, 0, ctx->stream>>>(d_syms, d_seq_byte_off, d_seq_len, d_seq_blk_off, n_seqs, total_blocks, d_planes); break;
    case 4: k_pack_planes<4><<<grid, block, 0, ctx->stream>>>(d_syms, d_seq_byte_off, d_seq_len, d_seq_blk_off, n_seqs, total_blocks, d_planes); break;
    case 8: k_pack_planes<8><<<grid, block, 0, ctx->stream>>>(d_syms, d_seq_byte_off, d_seq_len, d_seq_blk_off, n_seqs, total_blocks, d_planes); break;
    default: malcs_set_error("unsupported plane count %d", n_planes); return MALCS_ERR_ARG;
    }
    MALCS_HIP(hipGetLastError());
    return MALCS_OK;
}

// A caller's arena (malcs_batch_create_packed) may carry anything in the bits behind a sequence's last symbol; the
// library's own packers leave zeros there and the kernels that compare whole blocks count on it.  One thread per pair
// clears them in the library's copy (pairs that share a sequence write the same words with the same values).
__global__ void k_clear_tails(uint32_t *planes, const uint32_t *a_off, const uint32_t *a_len, const uint32_t *b_off,
                              const uint32_t *b_len, uint32_t n_pairs, int NP)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pairs) return;
    for (int side = 0; side < 2; side++) {
        const uint32_t off = side ? b_off[i] : a_off[i], len = side ? b_len[i] : a_len[i];
        if ((len & 31u) == 0) continue;
        const uint32_t keep = (1u << (len & 31u)) - 1u;
        uint32_t *last = planes + off + (uint64_t)(len >> 5) * NP;
        for (int p = 0; p < NP; p++) { const uint32_t w = last[p]; if (w & ~keep) last[p] = w & keep; }
    }
}

int malcs_clear_tails_launch(malcs_ctx *ctx, uint32_t *d_planes, const uint32_t *d_a_off, const uint32_t *d_a_len,
                             const uint32_t *d_b_off, const uint32_t *d_b_len, uint32_t n_pairs, int n_planes)
{
    if (n_pairs == 0) return MALCS_OK;
    k_clear_tails<<<dim3((n_pairs + 255u) / 256u), dim3(256), 0, ctx->stream>>>(d_planes, d_a_off, d_a_len, d_b_off, d_b_len, n_pairs, n_planes);
    MALCS_HIP(hipGetLastError());
    return MALCS_OK;
}
