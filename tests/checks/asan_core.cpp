// Sanitizer run of the host code of the graph path, on the CPU ("-fsanitize=address,undefined build of
// the CPU restatement"): the oracle (oracle/lcs_oracle.c: edit_distance, _lcs_nodes, _build_graph) and the library's own
// construction (algebra_amd/csrc/malcs_graph_core.hpp, the source the host half and the device kernel share) over a
// few thousand seeded pairs -- random, indel, tandem-repeat, homopolymer, empty sides, shifts -- compared with each other
// while AddressSanitizer and UBSan watch.  TEST INFRASTRUCTURE (it links the oracle).  Built and run by `make -C oracle asan`.
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
#include "../../algebra_amd/csrc/malcs_graph_core.hpp"

extern "C" {
struct orc_graph;
int64_t orc_edit_distance(const uint8_t *a, int64_t m, const uint8_t *b, int64_t n);
orc_graph *orc_graph_create(const uint8_t *reference, int64_t m, const uint8_t *observed, int64_t n, int64_t shift, int64_t max_distance);
void orc_graph_free(orc_graph *G);
int orc_graph_status(const orc_graph *G);
int64_t orc_graph_distance(const orc_graph *G);
int64_t orc_graph_n_levels(const orc_graph *G);
int64_t orc_graph_n_runs(const orc_graph *G);
int64_t orc_graph_n_nodes(const orc_graph *G);
int64_t orc_graph_n_edges(const orc_graph *G);
void orc_graph_supremal(const orc_graph *G, int64_t *out4);
void orc_graph_runs(const orc_graph *G, int64_t *out4);
void orc_graph_nodes(const orc_graph *G, int64_t *out3);
void orc_graph_edges(const orc_graph *G, int64_t *out6);
}

int main(int argc, char **argv)
{
    const int cases = argc > 1 ? atoi(argv[1]) : 3000;
    std::mt19937 rng(12345);
    auto rnd = [&](int n) { return (int)(rng() % (uint32_t)std::max(n, 1)); };
    int bad = 0, built = 0;
    for (int c = 0; c < cases; c++) {
        const int alphabet = 1 + rnd(4), len = rnd(3) == 0 ? rnd(12) : rnd(160);
        std::vector<uint8_t> a(len), b;
        for (auto &x : a) x = (uint8_t)rnd(alphabet);
        const int kind = rnd(5);
        if (kind == 0) { b.resize(rnd(160)); for (auto &x : b) x = (uint8_t)rnd(alphabet); }
        else if (kind == 1) b = a;
        else {
            for (uint8_t x : a) { if (rnd(12) == 0) continue; if (rnd(12) == 0) b.push_back((uint8_t)rnd(alphabet)); b.push_back(x); }
            if (kind == 3 && !a.empty()) { const int at = rnd((int)a.size()), unit = 1 + rnd(4); for (int r = 0; r < 3; r++) b.insert(b.begin() + std::min<size_t>(b.size(), at), a.begin() + at, a.begin() + std::min<int>((int)a.size(), at + unit)); }
            if (kind == 4) b.clear();
        }
        const int64_t shift = rnd(3) == 0 ? rnd(9) : 0, m = (int64_t)a.size(), n = (int64_t)b.size();
        orc_graph *G = orc_graph_create(a.data(), m, b.data(), n, shift, 0);
        if (!G || orc_graph_status(G) != 0) { printf("case %d: oracle failed\n", c); bad++; if (G) orc_graph_free(G); continue; }
        if (orc_graph_distance(G) != orc_edit_distance(a.data(), m, b.data(), n)) { printf("case %d: distance\n", c); bad++; }
        const int64_t R = orc_graph_n_runs(G), L = orc_graph_n_levels(G);
        std::vector<int64_t> runs(4 * std::max<int64_t>(R, 1)), sup(4), nodes(3 * std::max<int64_t>(orc_graph_n_nodes(G), 1)), edges(6 * std::max<int64_t>(orc_graph_n_edges(G), 1));
        orc_graph_runs(G, runs.data()); orc_graph_supremal(G, sup.data()); orc_graph_nodes(G, nodes.data()); orc_graph_edges(G, edges.data());
        if (R > 0 && L > 0) {
            // level lists = the runs grouped by level, in the order the sweep met them
            std::vector<int64_t> order(R);
            for (int64_t i = 0; i < R; i++) order[i] = i;
            std::stable_sort(order.begin(), order.end(), [&](int64_t x, int64_t y) { return runs[4 * x + 3] < runs[4 * y + 3]; });
            std::vector<int32_t> r32(4 * R);
            for (int64_t i = 0; i < R; i++) {
                const int64_t *x = &runs[4 * order[i]];
                r32[4 * i] = (int32_t)(x[0] - shift); r32[4 * i + 1] = (int32_t)(x[1] - shift); r32[4 * i + 2] = (int32_t)x[2]; r32[4 * i + 3] = (int32_t)x[3];
            }
            // small arrays on purpose, so that running out of room happens all the time: even cases hand the pair back and
            // retry with a larger arena (what the device does through the host), odd cases grow in the middle (the host)
            int32_t node_cap = 4, edge_cap = 4, list_cap = 2;
            for (;;) {
                std::vector<int32_t> node[9], edge, list[2];
                malcs_core::arena A;
                auto bind = [&]() {
                    int32_t **np[9] = {&A.row, &A.col, &A.length, &A.wlen, &A.incoming, &A.deg, &A.tail, &A.id, &A.order};
                    for (int k = 0; k < 9; k++) *np[k] = node[k].data();
                    A.edge = edge.data();
                    A.node_cap = (int32_t)node[0].size(); A.edge_cap = (int32_t)(edge.size() / 6); A.list_cap = (int32_t)list[0].size();
                };
                for (auto &v : node) v.assign((size_t)node_cap, 0);
                edge.assign(6 * (size_t)edge_cap, 0);
                for (auto &v : list) v.assign((size_t)list_cap, 0);
                bind();
                A.cur = list[0].data(); A.below = list[1].data();
                A.n_nodes = A.n_edges = A.n_cur = A.n_below = 0; A.overflow = 0;
                int32_t source = 0, sink = 0, s4[4];
                if (c % 2 == 0) {
                    malcs_core::build(A, (int32_t)m, (int32_t)n, shift, r32.data(), (int32_t)R, (int32_t)L, source, sink, s4, malcs_core::no_growth());
                } else {
                    malcs_core::build(A, (int32_t)m, (int32_t)n, shift, r32.data(), (int32_t)R, (int32_t)L, source, sink, s4,
                                      [&](malcs_core::arena &, int which) {
                                          if (which == malcs_core::GROW_NODES) for (auto &v : node) v.resize(v.size() * 2);
                                          else if (which == malcs_core::GROW_EDGES) edge.resize(edge.size() * 2);
                                          else {
                                              const bool first = A.cur == list[0].data();
                                              for (auto &v : list) v.resize(v.size() * 2);
                                              A.cur = list[first ? 0 : 1].data(); A.below = list[first ? 1 : 0].data();
                                          }
                                          bind();
                                          return true;
                                      });
                }
                if (A.overflow) { node_cap *= 2; edge_cap *= 2; list_cap *= 2; continue; }
                int64_t ne = 0;
                const int32_t nn = malcs_core::count_reachable(A, source, ne);
                std::vector<int64_t> gn(3 * (size_t)nn + 1), ge(6 * (size_t)ne + 1);
                std::vector<int32_t> where((size_t)A.n_edges + 1);                   // odd cases: with the table of written entries
                if (c % 3 == 2) {                                                    // in two parts, the copies in two halves out of order
                    std::vector<int32_t> at((size_t)nn + 1), pre((size_t)nn + 1), from((size_t)nn + 1);
                    malcs_core::write_graph_own<int64_t>(A, source, gn.data(), ge.data(), shift, where.data(), at.data(), pre.data(), from.data());
                    malcs_core::copy_prefixes<int64_t>(ge.data(), at.data(), pre.data(), from.data(), nn / 2, nn);
                    malcs_core::copy_prefixes<int64_t>(ge.data(), at.data(), pre.data(), from.data(), 0, nn / 2);
                } else
                malcs_core::write_graph<int64_t>(A, source, gn.data(), ge.data(), shift, c % 2 ? where.data() : nullptr);
                const bool same = nn == orc_graph_n_nodes(G) && ne == orc_graph_n_edges(G) && s4[0] + shift == sup[0] && s4[1] + shift == sup[1] &&
                                  s4[2] == sup[2] && s4[3] == sup[3] && !memcmp(gn.data(), nodes.data(), 3 * (size_t)nn * 8) &&
                                  !memcmp(ge.data(), edges.data(), 6 * (size_t)ne * 8);
                if (!same) { printf("case %d: graphs differ (m %lld n %lld shift %lld)\n", c, (long long)m, (long long)n, (long long)shift); bad++; }
                built++;
                break;
            }
        }
        orc_graph_free(G);
    }
    printf("%d cases, %d graphs built by both, %d mismatches\n", cases, built, bad);
    return bad ? 1 : 0;
}
