// Dev tool: times malcs_graph_core.hpp's build on level lists saved to a file (tools/time_host_build.py writes it).
//   g++ -O3 -std=c++17 -march=native -DMALCS_CORE_STATS -I algebra_amd/csrc tools/core_bench.cpp -o /tmp/cb/core_bench
#include "malcs_graph_core.hpp"
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
struct builder {
    std::vector<int32_t> node[9], edge, list[2];
    malcs_core::arena A;
    void bind()
    {
        int32_t **n[9] = {&A.row, &A.col, &A.length, &A.wlen, &A.incoming, &A.deg, &A.tail, &A.id, &A.order};
        for (int k = 0; k < 9; k++) *n[k] = node[k].data();
        A.edge = edge.data();
        A.node_cap = (int32_t)node[0].size(); A.edge_cap = (int32_t)(edge.size() / 6); A.list_cap = (int32_t)list[0].size();
    }
    void start(int32_t nodes, int32_t edges, int32_t lists)
    {
        for (auto &v : node) if ((int32_t)v.size() < nodes) v.resize(nodes);
        if (edge.size() < 6 * (size_t)edges) edge.resize(6 * (size_t)edges);
        for (auto &v : list) if ((int32_t)v.size() < lists) v.resize(lists);
        for (auto &v : node) v.resize(node[0].size());
        for (auto &v : list) v.resize(list[0].size());
        bind();
        A.cur = list[0].data(); A.below = list[1].data();
        A.n_nodes = A.n_edges = A.n_cur = A.n_below = 0; A.overflow = 0;
    }
    bool grow(int which)
    {
        auto wider = [](std::vector<int32_t> &v) { v.resize(v.size() * 2); return true; };
        if (which == malcs_core::GROW_NODES) { for (auto &v : node) wider(v); }
        else if (which == malcs_core::GROW_EDGES) { wider(edge); }
        else {
            const bool cur_is_0 = A.cur == list[0].data();
            for (auto &v : list) wider(v);
            A.cur = list[cur_is_0 ? 0 : 1].data(); A.below = list[cur_is_0 ? 1 : 0].data();
        }
        bind();
        return true;
    }
};
#ifdef MALCS_CORE_STATS
namespace malcs_core { long long stat_tests = 0, stat_shift = 0, stat_levels = 0; }
#endif
int main(int argc, char **argv)
{
    FILE *f = fopen(argc > 1 ? argv[1] : "/tmp/cb/c3.bin", "rb");
    int32_t count; if (fread(&count, 4, 1, f) != 1) return 1;
    builder B;
    for (int g = 0; g < count; g++) {
        int32_t h[4]; if (fread(h, 4, 4, f) != 4) return 1;
        std::vector<int32_t> runs((size_t)h[3] * 4); if (fread(runs.data(), 4, runs.size(), f) != runs.size()) return 1;
        double best = 1e9, bw = 1e9; int64_t ne = 0; int32_t nn = 0; unsigned long long sum = 0;
        std::vector<int32_t> nodes, edges, where;
        double bc = 1e9;
        for (int rep = 0; rep < 3; rep++) {
#ifdef MALCS_CORE_STATS
            malcs_core::stat_tests = malcs_core::stat_shift = malcs_core::stat_levels = 0;
#endif
            auto t0 = std::chrono::steady_clock::now();
            B.start(2 * h[3] + 16, 4 * h[3] + 64, h[3] + 8);
            int32_t source = 0, sink = 0, sup[4];
            malcs_core::build(B.A, h[0], h[1], 0, runs.data(), h[3], h[2], source, sink, sup, [&B](malcs_core::arena &, int w) { return B.grow(w); });
            auto t1 = std::chrono::steady_clock::now();
            nn = malcs_core::count_reachable(B.A, source, ne);
            if (nodes.size() < (size_t)nn * 3) nodes.resize((size_t)nn * 3);
            if (edges.size() < (size_t)ne * 6) edges.resize((size_t)ne * 6);
            auto t1b = std::chrono::steady_clock::now();
            where.resize((size_t)B.A.n_edges);
            malcs_core::write_graph(B.A, source, nodes.data(), edges.data(), 0, getenv("NOWHERE") ? nullptr : where.data());
            auto t2 = std::chrono::steady_clock::now();
            sum = 0; for (size_t k = 0; k < (size_t)ne * 6; k++) sum = sum * 1000003ull + (uint32_t)edges[k]; for (size_t k = 0; k < (size_t)nn * 3; k++) sum = sum * 1000003ull + (uint32_t)nodes[k];
            best = std::min(best, std::chrono::duration<double, std::milli>(t1 - t0).count());
            bw = std::min(bw, std::chrono::duration<double, std::milli>(t2 - t1b).count());
            bc = std::min(bc, std::chrono::duration<double, std::milli>(t1b - t1).count());
        }
        printf("graph %2d: runs %6d levels %5d -> nodes %5d edges %7lld  build %6.2f ms  count %5.2f ms  write %5.2f ms  checksum %016llx", g, h[3], h[2], nn, (long long)ne, best, bc, bw, sum);
#ifdef MALCS_CORE_STATS
        printf("  tests %lld shifted %lld levels %lld", malcs_core::stat_tests, malcs_core::stat_shift, malcs_core::stat_levels);
#endif
        printf("\n");
    }
    return 0;
}
