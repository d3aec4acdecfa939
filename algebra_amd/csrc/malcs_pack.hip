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
