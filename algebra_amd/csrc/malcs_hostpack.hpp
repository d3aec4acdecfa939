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

// Worker threads of a context, started once and kept: a batch runs several short parallel stretches in a row (places in
// the arena, packing, classes), and starting sixteen threads costs what the first of those stretches takes.  A worker
// that has finished spins for a moment before it goes to sleep, so that the next stretch of the same call finds it awake.
class malcs_workers {
public:
    explicit malcs_workers(int n_threads);          // n_threads - 1 threads are started; the caller is the n-th
    ~malcs_workers();
    malcs_workers(const malcs_workers &) = delete;
    malcs_workers &operator=(const malcs_workers &) = delete;
    int size() const { return n_; }
    // fn(k) for k = 1 ... use - 1 on worker threads (use <= size()); returns at once.  The caller does its own share
    // (k = 0, by convention) and then calls end(), which returns when every fn has.  One stretch at a time.
    void begin(int use, const std::function<void(int)> &fn);
    void end();
    // begin + fn(0) + end
    void run(int use, const std::function<void(int)> &fn) { begin(use, fn); fn(0); end(); }
private:
    struct impl;
    impl *p_;
    int n_;
};

// Packs all sequences with up to `max_threads` threads (`workers`: taken from there when given, else started and joined
// by the call).  `ship(first_dword, n_dwords)` is called on the calling thread
// for consecutive finished stretches of the arena, in order (return non-zero to report a failure: it becomes the
// result).  *codes_seen = OR of every code read (callers check it against the alphabet they were promised).
int malcs_host_pack(const malcs_pack_job &job, int max_threads, const std::function<int(size_t, size_t)> &ship, uint32_t *codes_seen,
                    malcs_workers *workers = nullptr);
