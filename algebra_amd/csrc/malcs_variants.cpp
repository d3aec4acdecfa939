// malcs_variants.cpp -- LCSgraph.from_variant for many alleles of one reference, inside the library.
//
// Stands in for the loop of algebra/lcs/lcs_graph.py:77-123 (`from_variant`): the allele's delins is put into a window
// of the reference that reaches `offset` symbols beyond it on both sides, the supremal LCS graph of (reference window,
// window with the allele) is built, and the window is doubled until the graph's supremal variant lies strictly inside
// it (or reaches an end of the reference) on both sides (:113-123).  The alleles of a batch are independent: every
// round is ONE device batch over the windows of all alleles that are still widening (malcs_batch_create +
// malcs_lcs_graph_batch on the context's stream), the windows are cut from the reference's codes here, and what the
// relation matrix reads of every graph (interval, distance, a number per distinct supremal, the symbols of its inserted
// sequence, its atom masks) is put together here as well -- the caller gets arrays, not 5 000 objects.
//
// An allele that changes nothing (reference[start:end] == inserted) is the graph of two empty sequences, as
// `from_variant` returns LCSgraph("", "") for it (:101-103); it rides in the first round's batch.
#include "malcs_internal.hpp"

#include <algorithm>
#include <cstring>
#include <unordered_map>
#include <vector>

struct malcs_vgraphs {
    malcs_ctx *ctx = nullptr;
    uint64_t n = 0;
    std::vector<malcs_graphs *> rounds;                  // one handle per widening round
    std::vector<uint64_t> round_size;
    std::vector<int64_t> where;                          // per allele: round, index in the round's handle, lo, hi of the final window
    std::vector<int64_t> distance, supremal;             // per allele; supremal: start, end, obs_from, obs_to
    std::vector<uint64_t> obs_off;                       // the observed windows (codes) of all alleles, back to back
    std::vector<uint8_t> obs;
    ~malcs_vgraphs() { for (malcs_graphs *g : rounds) malcs_graphs_destroy(g); }
};

extern "C" {

int malcs_from_variant_batch(malcs_ctx *ctx, const uint8_t *ref, uint64_t ref_len, uint64_t n, const int64_t *start,
                             const int64_t *end, const uint8_t *ins, const uint64_t *ins_off, int64_t offset, int n_symbols,
                             malcs_vgraphs **out)
{
    if (!out) { malcs_set_error("graphs is NULL"); return MALCS_ERR_ARG; }
    *out = nullptr;
    if (!ctx) { malcs_set_error("ctx is NULL"); return MALCS_ERR_ARG; }
    if ((ref_len && !ref) || (n && (!start || !end || !ins_off))) { malcs_set_error("an argument is NULL"); return MALCS_ERR_ARG; }
    if (n_symbols < 1 || n_symbols > 256) { malcs_set_error("n_symbols %d outside [1,256]", n_symbols); return MALCS_ERR_ARG; }
    for (uint64_t i = 0; i < n; i++) {
        if (start[i] < 0 || end[i] < start[i] || (uint64_t)end[i] > ref_len || ins_off[i + 1] < ins_off[i]) {
            malcs_set_error("allele %llu: [%lld, %lld) is not an interval of the reference (%llu symbols)", (unsigned long long)i,
                            (long long)start[i], (long long)end[i], (unsigned long long)ref_len);
            return MALCS_ERR_ARG;
        }
    }
    malcs_vgraphs *V = new (std::nothrow) malcs_vgraphs();
    if (!V) { malcs_set_error("out of host memory"); return MALCS_ERR_NOMEM; }
    V->ctx = ctx; V->n = n;
    V->where.assign(4 * n, 0); V->distance.assign(n, 0); V->supremal.assign(4 * n, 0);
    std::vector<std::vector<uint8_t>> round_syms;          // the symbol buffer of every round: the final windows are copied out at the end
    std::vector<uint64_t> obs_at(n, 0), obs_len(n, 0);

    struct item { uint64_t idx; int64_t off; bool trivial; };
    std::vector<item> pending;
    pending.reserve(n);
    for (uint64_t i = 0; i < n; i++) {
        const uint64_t li = ins_off[i + 1] - ins_off[i], ld = (uint64_t)(end[i] - start[i]);
        const bool trivial = li == ld && (li == 0 || !memcmp(ref + start[i], ins + ins_off[i], li));      // :101-103
        const int64_t len = (int64_t)(ld + li);                                                             // len(Variant), variant.py:77-78
        pending.push_back({i, std::max<int64_t>(std::max<int64_t>(offset, len / 2), 1), trivial});       // :111
    }
    std::vector<uint8_t> syms;
    std::vector<uint64_t> a_off, b_off;
    std::vector<uint32_t> a_len, b_len;
    std::vector<int64_t> shift, lo_of, hi_of, dist, sup;
    std::vector<int32_t> status;
    std::vector<int64_t> counts;
    int rc = MALCS_OK;
    while (!pending.empty() && rc == MALCS_OK) {
        const uint64_t np = pending.size();
        // the reference once, then the windows with the alleles in them; a pair's reference window is a stretch of the first part
        uint64_t total = ref_len;
        for (const item &p : pending) {
            if (p.trivial) continue;
            const int64_t lo = std::max<int64_t>(0, start[p.idx] - p.off), hi = std::min<int64_t>((int64_t)ref_len, end[p.idx] + p.off);
            total += (uint64_t)(start[p.idx] - lo) + (ins_off[p.idx + 1] - ins_off[p.idx]) + (uint64_t)(hi - end[p.idx]);
        }
        syms.resize(total + 1);
        if (ref_len) memcpy(syms.data(), ref, ref_len);
        a_off.resize(np); b_off.resize(np); a_len.resize(np); b_len.resize(np); shift.resize(np); lo_of.resize(np); hi_of.resize(np);
        uint64_t at = ref_len;
        bool too_long = false;
        for (uint64_t k = 0; k < np; k++) {
            const item &p = pending[k];
            if (p.trivial) { a_off[k] = b_off[k] = 0; a_len[k] = b_len[k] = 0; shift[k] = 0; lo_of[k] = hi_of[k] = 0; continue; }
            const uint64_t i = p.idx;
            const int64_t lo = std::max<int64_t>(0, start[i] - p.off), hi = std::min<int64_t>((int64_t)ref_len, end[i] + p.off);   // :114-115
            const uint64_t li = ins_off[i + 1] - ins_off[i];
            const uint64_t wlen = (uint64_t)(start[i] - lo) + li + (uint64_t)(hi - end[i]);
            too_long = too_long || wlen > 0xffffffffull || (uint64_t)(hi - lo) > 0xffffffffull;
            a_off[k] = (uint64_t)lo; a_len[k] = (uint32_t)(hi - lo);
            b_off[k] = at; b_len[k] = (uint32_t)wlen;
            uint8_t *w = syms.data() + at;                                   // :116-118
            memcpy(w, ref + lo, (size_t)(start[i] - lo)); w += start[i] - lo;
            if (li) memcpy(w, ins + ins_off[i], li);
            w += li;
            memcpy(w, ref + end[i], (size_t)(hi - end[i]));
            at += wlen;
            shift[k] = lo; lo_of[k] = lo; hi_of[k] = hi;
        }
        if (too_long) { malcs_set_error("a window beyond 2^32 symbols"); rc = MALCS_ERR_TOO_LONG; break; }
        malcs_batch *b = nullptr;
        rc = malcs_batch_create(ctx, syms.data(), total, a_off.data(), a_len.data(), b_off.data(), b_len.data(), np, n_symbols, &b);
        if (rc) break;
        malcs_graphs *g = nullptr;
        rc = malcs_lcs_graph_batch(b, shift.data(), nullptr, &g);            // :119
        malcs_batch_destroy(b);
        if (rc) break;
        const int round = (int)V->rounds.size();
        V->rounds.push_back(g);
        V->round_size.push_back(np);
        counts.resize(4 * np); status.resize(np); dist.resize(np); sup.resize(4 * np);
        rc = malcs_graphs_columns(g, counts.data(), status.data(), dist.data(), sup.data());
        if (rc) break;
        std::vector<item> again;
        for (uint64_t k = 0; k < np; k++) {
            const item &p = pending[k];
            const int64_t s0 = sup[4 * k], s1 = sup[4 * k + 1];
            const bool inside_left = s0 > lo_of[k] || s0 == 0;                              // :120
            const bool inside_right = s1 < hi_of[k] || s1 == (int64_t)ref_len;             // :121
            if (p.trivial || (inside_left && inside_right)) {
                const uint64_t i = p.idx;
                V->where[4 * i] = round; V->where[4 * i + 1] = (int64_t)k; V->where[4 * i + 2] = lo_of[k]; V->where[4 * i + 3] = hi_of[k];
                V->distance[i] = dist[k];
                for (int c = 0; c < 4; c++) V->supremal[4 * i + c] = sup[4 * k + c];
                obs_at[i] = b_off[k]; obs_len[i] = b_len[k];
            } else {
                again.push_back({p.idx, p.off * 2, false});                                 // :123
            }
        }
        pending.swap(again);
        round_syms.emplace_back(std::move(syms));
        syms.clear();
    }
    if (rc) { delete V; return rc; }
    V->obs_off.assign(n + 1, 0);
    for (uint64_t i = 0; i < n; i++) V->obs_off[i + 1] = V->obs_off[i] + obs_len[i];
    V->obs.resize(V->obs_off[n] + 1);
    for (uint64_t i = 0; i < n; i++)
        if (obs_len[i]) memcpy(V->obs.data() + V->obs_off[i], round_syms[(size_t)V->where[4 * i]].data() + obs_at[i], obs_len[i]);
    *out = V;
    return MALCS_OK;
}

int malcs_vgraphs_destroy(malcs_vgraphs *v) { delete v; return MALCS_OK; }

uint64_t malcs_vgraphs_size(const malcs_vgraphs *v) { return v ? v->n : 0; }

int malcs_vgraphs_columns(const malcs_vgraphs *v, int64_t *where4, int64_t *distance, int64_t *supremal4, uint64_t *obs_off)
{
    if (!v) { malcs_set_error("graphs is NULL"); return MALCS_ERR_ARG; }
    if (where4) memcpy(where4, v->where.data(), v->where.size() * sizeof(int64_t));
    if (distance) memcpy(distance, v->distance.data(), v->distance.size() * sizeof(int64_t));
    if (supremal4) memcpy(supremal4, v->supremal.data(), v->supremal.size() * sizeof(int64_t));
    if (obs_off) memcpy(obs_off, v->obs_off.data(), v->obs_off.size() * sizeof(uint64_t));
    return MALCS_OK;
}

int malcs_vgraphs_observed(const malcs_vgraphs *v, uint8_t *obs)
{
    if (!v || !obs) { malcs_set_error("an argument is NULL"); return MALCS_ERR_ARG; }
    memcpy(obs, v->obs.data(), v->obs_off[v->n]);
    return MALCS_OK;
}

int malcs_vgraphs_rounds(const malcs_vgraphs *v, int *n_rounds)
{
    if (!v || !n_rounds) { malcs_set_error("an argument is NULL"); return MALCS_ERR_ARG; }
    *n_rounds = (int)v->rounds.size();
    return MALCS_OK;
}

int malcs_vgraphs_round(const malcs_vgraphs *v, int round, malcs_graphs **graphs, uint64_t *n_pairs)
{
    if (!v || !graphs || round < 0 || round >= (int)v->rounds.size()) { malcs_set_error("no such round"); return MALCS_ERR_ARG; }
    *graphs = v->rounds[(size_t)round];
    if (n_pairs) *n_pairs = v->round_size[(size_t)round];
    return MALCS_OK;
}

int malcs_vgraphs_matrix_columns(malcs_vgraphs *v, uint32_t *key, uint64_t *symmask, int *folded, const uint64_t *sup_off, uint8_t *sup,
                                 const uint64_t *atom_off, uint8_t *atoms)
{
    if (!v || !key || !symmask || !folded || !sup_off) { malcs_set_error("an argument is NULL"); return MALCS_ERR_ARG; }
    const uint64_t n = v->n;
    // a number per distinct supremal (start, end, sequence): equal supremals <=> equal numbers (graph_based.py:51);
    // one bit per symbol of its sequence (Variant.is_disjoint, variant.py:133), folded beyond 64 symbols
    std::unordered_multimap<uint64_t, uint32_t> seen;
    seen.reserve((size_t)n * 2);
    uint32_t next = 0;
    int any_high = 0;
    for (uint64_t i = 0; i < n; i++) {
        const int64_t *s = &v->supremal[4 * i];
        const uint8_t *seq = v->obs.data() + v->obs_off[i] + s[2];
        const uint64_t len = (uint64_t)(s[3] - s[2]);
        if (sup_off[i + 1] - sup_off[i] != len) { malcs_set_error("sup_off does not match the supremals"); return MALCS_ERR_ARG; }
        if (sup && len) memcpy(sup + sup_off[i], seq, len);
        uint64_t h = 0xcbf29ce484222325ull ^ (uint64_t)s[0] * 0x9e3779b97f4a7c15ull ^ (uint64_t)s[1] * 0xc2b2ae3d27d4eb4full, mask = 0;
        for (uint64_t k = 0; k < len; k++) { h = (h ^ seq[k]) * 0x100000001b3ull; mask |= 1ull << (seq[k] & 63u); any_high |= seq[k] >= 64; }
        symmask[i] = mask;
        uint32_t id = 0xffffffffu;
        auto range = seen.equal_range(h);
        for (auto it = range.first; it != range.second; ++it) {
            const uint64_t j = it->second;
            const int64_t *t = &v->supremal[4 * j];
            if (t[0] == s[0] && t[1] == s[1] && (uint64_t)(t[3] - t[2]) == len &&
                (len == 0 || !memcmp(v->obs.data() + v->obs_off[j] + t[2], seq, len))) { id = key[j]; break; }
        }
        if (id == 0xffffffffu) { id = next++; seen.emplace(h, (uint32_t)i); }
        key[i] = id;
    }
    *folded = any_high;
    if (atoms && atom_off) {
        // the atom masks, round by round (malcs_graphs_atoms works on one handle)
        std::vector<uint64_t> index, off, range;
        std::vector<uint8_t> obs;
        for (size_t r = 0; r < v->rounds.size(); r++) {
            index.clear(); off.assign(1, 0); range.clear(); obs.clear();
            for (uint64_t i = 0; i < n; i++) {
                if (v->where[4 * i] != (int64_t)r) continue;
                index.push_back((uint64_t)v->where[4 * i + 1]);
                obs.insert(obs.end(), v->obs.begin() + (ptrdiff_t)v->obs_off[i], v->obs.begin() + (ptrdiff_t)v->obs_off[i + 1]);
                off.push_back(obs.size());
                range.push_back(atom_off[i]); range.push_back(atom_off[i + 1]);
            }
            if (index.empty()) continue;
            obs.push_back(0);
            const int rc = malcs_graphs_atoms(v->rounds[r], index.size(), index.data(), obs.data(), off.data(), range.data(), atoms);
            if (rc) return rc;
        }
    }
    return MALCS_OK;
}

}  // extern "C"
