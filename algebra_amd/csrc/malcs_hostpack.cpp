// malcs_hostpack.cpp -- host side of batch creation: uint8 symbol codes -> bit planes, on the host's cores.
//
// A batch arrives as bytes (one symbol each); the device wants NP bits per symbol.  Packing on the host before the
// copy means a quarter (DNA) of the bytes cross PCIe, and the packing itself is a streaming pass at memory speed:
// 32 symbols -> NP dwords with one shift + one byte-mask-to-bits instruction per plane (AVX2 vpmovmskb; a multiply
// that gathers one bit of each of 8 bytes where AVX2 is missing).  Worker threads take chunks of sequences from a
// counter; the calling thread hands every finished chunk, in order, to `ship` (malcs_api.cpp queues its copy to the
// device there), so that packing and copying overlap.
//
// Layout written: the packed sequence arena (malcs_pack.hip): per sequence, per block of 32 symbols, NP dwords; bit j of
// plane p = bit p of the code of symbol 32 k + j; symbols past the end of a sequence pack as code 0.
#include "malcs_hostpack.hpp"
#include <atomic>
#include <algorithm>
#include <cstring>
#include <thread>
#include <mutex>
#include <condition_variable>
#include <system_error>
#include <immintrin.h>

namespace {

// plane p of 8 symbols: bit j = bit p of byte j
inline uint32_t gather8(uint64_t x, int p)
{
    return (uint32_t)((((x >> p) & 0x0101010101010101ull) * 0x0102040810204080ull) >> 56);
}

template <int NP>
void pack_block_scalar(const uint8_t *src, uint32_t cnt, uint32_t *out, uint64_t &seen)
{
    uint64_t w[4] = {0, 0, 0, 0};
    if (cnt >= 32) memcpy(w, src, 32); else memcpy(w, src, cnt);
    seen |= w[0] | w[1] | w[2] | w[3];
    for (int p = 0; p < NP; p++)
        out[p] = gather8(w[0], p) | (gather8(w[1], p) << 8) | (gather8(w[2], p) << 16) | (gather8(w[3], p) << 24);
}

template <int NP>
void pack_range_scalar(const malcs_pack_job &job, uint32_t s0, uint32_t s1, uint64_t &seen_out)
{
    uint64_t seen = 0;
    for (uint32_t s = s0; s < s1; s++) {
        const uint8_t *src = job.syms + job.seq_byte_off[s];
        const uint32_t len = job.seq_len[s];
        uint32_t *out = job.out + (size_t)job.seq_blk_off[s] * NP;
        for (uint32_t k = 0; k * 32u < len; k++) pack_block_scalar<NP>(src + (size_t)k * 32u, std::min(32u, len - k * 32u), out + (size_t)k * NP, seen);
    }
    seen_out |= seen;
}

template <int NP>
__attribute__((target("avx2"))) void pack_range_avx2(const malcs_pack_job &job, uint32_t s0, uint32_t s1, uint64_t &seen_out)
{
    __m256i seen = _mm256_setzero_si256();
    for (uint32_t s = s0; s < s1; s++) {
        const uint8_t *src = job.syms + job.seq_byte_off[s];
        const uint32_t len = job.seq_len[s];
        uint32_t *out = job.out + (size_t)job.seq_blk_off[s] * NP;
        const uint32_t whole = len / 32u;
        for (uint32_t k = 0; k < whole; k++) {
            if ((k & 1u) == 0) _mm_prefetch(reinterpret_cast<const char *>(src + (size_t)k * 32u + 1024), _MM_HINT_NTA);   // a line, 1 KB ahead (1.43 -> 1.33 ms for the 200 MB of a C2 batch)
            const __m256i x = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(src + (size_t)k * 32u));
            seen = _mm256_or_si256(seen, x);
            // bit p of every byte to the byte's top bit (a 16-bit shift leaks bits of the lower byte into the upper one,
            // but only below its top bit), then one bit per byte
            if (NP > 0) out[(size_t)k * NP + 0] = (uint32_t)_mm256_movemask_epi8(_mm256_slli_epi16(x, 7));
            if (NP > 1) out[(size_t)k * NP + 1] = (uint32_t)_mm256_movemask_epi8(_mm256_slli_epi16(x, 6));
            if (NP > 2) out[(size_t)k * NP + 2] = (uint32_t)_mm256_movemask_epi8(_mm256_slli_epi16(x, 5));
            if (NP > 3) out[(size_t)k * NP + 3] = (uint32_t)_mm256_movemask_epi8(_mm256_slli_epi16(x, 4));
            if (NP > 4) {
                out[(size_t)k * NP + 4] = (uint32_t)_mm256_movemask_epi8(_mm256_slli_epi16(x, 3));
                out[(size_t)k * NP + 5] = (uint32_t)_mm256_movemask_epi8(_mm256_slli_epi16(x, 2));
                out[(size_t)k * NP + 6] = (uint32_t)_mm256_movemask_epi8(_mm256_slli_epi16(x, 1));
                out[(size_t)k * NP + 7] = (uint32_t)_mm256_movemask_epi8(x);
            }
        }
        if (len % 32u) {
            uint64_t tail_seen = 0;
            pack_block_scalar<NP>(src + (size_t)whole * 32u, len % 32u, out + (size_t)whole * NP, tail_seen);
            seen = _mm256_or_si256(seen, _mm256_set1_epi64x((long long)tail_seen));
        }
    }
    alignas(32) uint64_t lanes[4];
    _mm256_store_si256(reinterpret_cast<__m256i *>(lanes), seen);
    seen_out |= lanes[0] | lanes[1] | lanes[2] | lanes[3];
}

typedef void (*range_fn)(const malcs_pack_job &, uint32_t, uint32_t, uint64_t &);

range_fn pick(int np)
{
    const bool avx2 = __builtin_cpu_supports("avx2");
    switch (np) {
    case 2: return avx2 ? pack_range_avx2<2> : pack_range_scalar<2>;
    case 3: return avx2 ? pack_range_avx2<3> : pack_range_scalar<3>;
    case 4: return avx2 ? pack_range_avx2<4> : pack_range_scalar<4>;
    case 8: return avx2 ? pack_range_avx2<8> : pack_range_scalar<8>;
    default: return nullptr;
    }
}

}  // namespace

struct malcs_workers::impl {
    std::mutex m;
    std::condition_variable wake, done;
    std::vector<std::thread> threads;
    const std::function<void(int)> *job = nullptr;
    std::atomic<uint64_t> generation{0};
    int use = 0;                     // workers 1 ... use - 1 take part in the current stretch
    std::atomic<int> running{0};
    bool stop = false;

    void loop(int k)
    {
        uint64_t seen = 0;
        for (;;) {
            // awake for a moment after a stretch: the next one of the same call comes within microseconds
            for (int spin = 0; spin < 20000 && generation.load(std::memory_order_acquire) == seen; spin++) _mm_pause();
            const std::function<void(int)> *fn = nullptr;
            {
                std::unique_lock<std::mutex> lock(m);
                wake.wait(lock, [&] { return stop || generation.load(std::memory_order_relaxed) != seen; });
                if (stop) return;
                seen = generation.load(std::memory_order_relaxed);
                if (k < use) fn = job;
            }
            if (!fn) continue;
            (*fn)(k);
            if (running.fetch_sub(1, std::memory_order_acq_rel) == 1) {
                std::lock_guard<std::mutex> lock(m);
                done.notify_all();
            }
        }
    }
};

malcs_workers::malcs_workers(int n_threads) : p_(new impl), n_(1)
{
    try {
        for (int k = 1; k < n_threads; k++) { p_->threads.emplace_back([this, k] { p_->loop(k); }); n_ = k + 1; }
    } catch (const std::system_error &) {}              // no more threads to be had: the ones we have do the work
}

malcs_workers::~malcs_workers()
{
    {
        std::lock_guard<std::mutex> lock(p_->m);
        p_->stop = true;
    }
    p_->wake.notify_all();
    for (std::thread &t : p_->threads) t.join();
    delete p_;
}

void malcs_workers::begin(int use, const std::function<void(int)> &fn)
{
    use = std::max(1, std::min(use, n_));
    {
        std::lock_guard<std::mutex> lock(p_->m);
        p_->job = &fn;
        p_->use = use;
        p_->running.store(use - 1, std::memory_order_relaxed);
        p_->generation.fetch_add(1, std::memory_order_release);
    }
    p_->wake.notify_all();
}

void malcs_workers::end()
{
    for (int spin = 0; spin < 20000 && p_->running.load(std::memory_order_acquire) != 0; spin++) _mm_pause();
    std::unique_lock<std::mutex> lock(p_->m);
    p_->done.wait(lock, [&] { return p_->running.load(std::memory_order_acquire) == 0; });
    p_->job = nullptr;
}

int malcs_host_pack(const malcs_pack_job &job, int max_threads, const std::function<int(size_t, size_t)> &ship, uint32_t *codes_seen,
                    malcs_workers *workers)
{
    range_fn fn = pick(job.n_planes);
    if (!fn) return -1;
    const uint32_t n = job.n_seqs;
    const uint64_t total_blocks = n ? (uint64_t)job.seq_blk_off[n - 1] + (job.seq_len[n - 1] + 31u) / 32u : 0;
    // chunks of about 32 K blocks (1 MB of symbols), cut at sequence boundaries
    std::vector<uint32_t> cut{0};
    {
        const uint64_t per = std::max<uint64_t>(32768, total_blocks / 256 + 1);
        uint64_t next = per;
        for (uint32_t s = 0; s < n; s++)
            if (job.seq_blk_off[s] >= next) { cut.push_back(s); next = job.seq_blk_off[s] + per; }
        cut.push_back(n);
    }
    const size_t n_chunks = cut.size() - 1;
    std::vector<std::atomic<uint8_t>> done(n_chunks);
    for (auto &d : done) d.store(0, std::memory_order_relaxed);
    std::atomic<size_t> next_chunk{0};
    std::atomic<uint64_t> seen_all{0};
    auto work = [&]() {
        uint64_t seen = 0;
        for (;;) {
            const size_t c = next_chunk.fetch_add(1, std::memory_order_relaxed);
            if (c >= n_chunks) break;
            fn(job, cut[c], cut[c + 1], seen);
            done[c].store(1, std::memory_order_release);
        }
        seen_all.fetch_or(seen, std::memory_order_relaxed);
    };
    const size_t want = std::min<size_t>((size_t)std::max(1, max_threads), n_chunks);
    std::vector<std::thread> pool;
    const std::function<void(int)> work_k = [&](int) { work(); };
    const bool kept = workers && workers->size() > 1 && want > 1;
    if (kept) {
        workers->begin((int)want, work_k);
    } else {
        try {
            for (size_t t = 1; t < want; t++) pool.emplace_back(work);
        } catch (const std::system_error &) {}          // no more threads to be had: the ones we have do the rest
    }
    int rc = 0;
    if (!kept && pool.empty()) work();                   // small input, or no threads: pack here, then ship
    // finished chunks leave in order, a few megabytes at a time (a copy has a fixed cost of several microseconds)
    const size_t ship_dwords = (4u << 20) / sizeof(uint32_t);
    size_t pending_first = 0, pending_end = 0;
    for (size_t c = 0; c < n_chunks; c++) {
        // with workers: this thread ships finished chunks and packs while it would otherwise wait
        while (!done[c].load(std::memory_order_acquire)) {
            const size_t mine = next_chunk.fetch_add(1, std::memory_order_relaxed);
            if (mine < n_chunks) {
                uint64_t seen = 0;
                fn(job, cut[mine], cut[mine + 1], seen);
                seen_all.fetch_or(seen, std::memory_order_relaxed);
                done[mine].store(1, std::memory_order_release);
            } else {
                std::this_thread::yield();
            }
        }
        pending_end = cut[c + 1] < n ? (size_t)job.seq_blk_off[cut[c + 1]] * job.n_planes : (size_t)total_blocks * job.n_planes;
        if (rc == 0 && (pending_end - pending_first >= ship_dwords || c + 1 == n_chunks) && pending_end > pending_first) {
            rc = ship(pending_first, pending_end - pending_first);
            pending_first = pending_end;
        }
    }
    if (kept) workers->end();
    for (std::thread &t : pool) t.join();
    uint64_t seen = seen_all.load();
    seen |= seen >> 32; seen |= seen >> 16; seen |= seen >> 8;
    if (codes_seen) *codes_seen = (uint32_t)(seen & 0xffu);
    return rc;
}
