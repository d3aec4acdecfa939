// malcs_hostpack.hpp -- host-side packing of symbol bytes into bit planes (malcs_hostpack.cpp; plain C++, no HIP).
#pragma once
#include <cstddef>
#include <cstdint>
#include <functional>
#include <vector>

struct malcs_pack_job {
    const uint8_t *syms;             // the caller's symbol buffer
    const uint64_t *seq_byte_off;    // per sequence: first byte ...
    const uint32_t *seq_len;         // ... symbols ...
    const uint32_t *seq_blk_off;     // ... and first block of 32 symbols in the arena (ascending: sequences back to back)
    uint32_t n_seqs;
    int n_planes;                    // 2, 3, 4 or 8
    uint32_t *out;                   // the arena, n_planes dwords per block (host memory, ideally pinned)
};

// Packs all sequences with up to `max_threads` threads.  `ship(first_dword, n_dwords)` is called on the calling thread
// for consecutive finished stretches of the arena, in order (return non-zero to report a failure: it becomes the
// result).  *codes_seen = OR of every code read (callers check it against the alphabet they were promised).
int malcs_host_pack(const malcs_pack_job &job, int max_threads, const std::function<int(size_t, size_t)> &ship, uint32_t *codes_seen);
