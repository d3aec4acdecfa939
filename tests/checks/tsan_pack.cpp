// ThreadSanitizer run of the library's host threads (algebra_amd/csrc/malcs_hostpack.cpp: malcs_workers and the packing
// of symbol bytes into bit planes on them), no device: many short parallel stretches in a row on one kept set of threads,
// packing with kept threads, with threads of its own and on one thread compared word by word.  Test infrastructure
// (make -C oracle tsan).
#include "../../algebra_amd/csrc/malcs_hostpack.hpp"
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

int main(int argc, char **argv)
{
    const int rounds = argc > 1 ? atoi(argv[1]) : 200;
    malcs_workers wk(8);
    // 1. stretches: every index once, results visible to the caller after run()
    std::mt19937 rng(7);
    for (int r = 0; r < rounds; r++) {
        const int use = 1 + (int)(rng() % (unsigned)wk.size());
        std::vector<int> hit((size_t)wk.size(), 0);
        long sum = 0;
        std::vector<long> part((size_t)wk.size(), 0);
        wk.run(use, [&](int k) { hit[(size_t)k]++; for (int i = 0; i < 1000; i++) part[(size_t)k] += i ^ k; });
        for (int k = 0; k < wk.size(); k++) {
            if (hit[(size_t)k] != (k < use ? 1 : 0)) { printf("round %d: worker %d ran %d times (use %d)\n", r, k, hit[(size_t)k], use); return 1; }
            sum += part[(size_t)k];
        }
        if (sum == -1) return 2;
    }
    // 2. packing: kept threads / own threads / one thread
    for (int np : {2, 3, 4, 8}) {
        const uint32_t n_seqs = 3000;
        std::vector<uint64_t> off(n_seqs);
        std::vector<uint32_t> len(n_seqs), blk(n_seqs);
        uint64_t bytes = 0, blocks = 0;
        for (uint32_t s = 0; s < n_seqs; s++) { len[s] = 1 + rng() % 3000; off[s] = bytes; bytes += len[s]; blk[s] = (uint32_t)blocks; blocks += (len[s] + 31) / 32; }
        std::vector<uint8_t> syms(bytes);
        for (auto &x : syms) x = (uint8_t)(rng() % (1u << np));
        std::vector<uint32_t> a(blocks * np, 0xdeadbeefu), b(blocks * np, 0xdeadbeefu), c(blocks * np, 0xdeadbeefu);
        uint32_t ca = 0, cb = 0, cc = 0;
        size_t shipped = 0;
        auto ship = [&](size_t first, size_t count) -> int { if (first != shipped) return 1; shipped += count; return 0; };
        malcs_pack_job ja{syms.data(), off.data(), len.data(), blk.data(), n_seqs, np, a.data()};
        if (malcs_host_pack(ja, 8, ship, &ca, &wk) || shipped != a.size()) { printf("np %d: kept threads failed\n", np); return 3; }
        shipped = 0;
        malcs_pack_job jb{syms.data(), off.data(), len.data(), blk.data(), n_seqs, np, b.data()};
        if (malcs_host_pack(jb, 4, ship, &cb, nullptr) || shipped != b.size()) { printf("np %d: own threads failed\n", np); return 3; }
        shipped = 0;
        malcs_pack_job jc{syms.data(), off.data(), len.data(), blk.data(), n_seqs, np, c.data()};
        if (malcs_host_pack(jc, 1, ship, &cc, nullptr) || shipped != c.size()) { printf("np %d: one thread failed\n", np); return 3; }
        if (a != c || b != c || ca != cc || cb != cc) { printf("np %d: results differ\n", np); return 4; }
        // plane p of block k against the bytes
        for (uint32_t s = 0; s < n_seqs; s += 97)
            for (uint32_t j = 0; j < len[s]; j++)
                for (int p = 0; p < np; p++)
                    if (((c[((size_t)blk[s] + j / 32) * np + p] >> (j % 32)) & 1u) != ((syms[off[s] + j] >> p) & 1u)) { printf("np %d: bit wrong\n", np); return 5; }
    }
    printf("tsan_pack: %d stretches and 4 alphabets, no findings\n", rounds);
    return 0;
}
