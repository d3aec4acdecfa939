"""Generate the golden fixtures under tests/golden/ from the real reference.

Run in the build container only (the reference is mounted read-only at /root/reference and never
travels to the GPU box):

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python3 /root/repo/tests/golden/make_golden.py

Inputs come from (a) the parametrize lists of the reference's own tests, read by importing its
test modules, and (b) seeded generators in tests/gen.py.  Expected outputs are whatever the
imported reference returns.  The JSON files hold data only (inputs and captured outputs).
"""

import hashlib
import json
import os
import sys
import time

REFERENCE = os.environ.get("ALGEBRA_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REFERENCE)
sys.path.insert(0, os.path.join(HERE, ".."))

import pytest  # noqa: E402,F401  (the reference's test modules import it)
from algebra import Relation  # noqa: E402
from algebra.extractor import extract_sequence, to_hgvs  # noqa: E402
from algebra.extractor.extractor import canonical  # noqa: E402
from algebra.lcs import LCSgraph, edit_distance  # noqa: E402
from algebra.lcs.lcs_graph import _lcs_nodes, trim  # noqa: E402
from algebra.relations import graph_based, sequence_based, supremal_based, variant_based  # noqa: E402
from algebra.variants import Variant, patch  # noqa: E402

import gen  # noqa: E402  (tests/gen.py)


def digest(obj):
    return hashlib.sha256(json.dumps(obj, separators=(",", ":"), sort_keys=True).encode()).hexdigest()


def var(v):
    return [v.start, v.end, v.sequence]


def node(n):
    return [n.row, n.col, n.length]


def graph_record(graph):
    """Canonical form of an LCSgraph: BFS by identity from the source following edge order."""
    ids = {id(graph._source): 0}
    order = [graph._source]
    head = 0
    edges = []
    while head < len(order):
        src = order[head]
        for child, variant in src.edges:
            if id(child) not in ids:
                ids[id(child)] = len(order)
                order.append(child)
            edges.append([head, ids[id(child)]] + var(variant))
        head += 1
    return {
        "distance": graph.distance,
        "supremal": var(graph.supremal),
        "nodes": [node(n) for n in order],
        "edges": edges,
    }


def observables(graph):
    """What callers see through the public methods (order-sensitive)."""
    return {
        "nodes_order": [node(n) for n in graph.nodes()],
        "bfs": [[node(s), node(t)] + var(v[0]) for s, t, v in graph.bfs_traversal()],
        "edges_sorted": sorted(var(v) for v in graph.edges()),
        "uniq_atomics": sorted([i, ch] for i, ch in graph.uniq_atomics()),
    }


def lcs_case(reference, observed, shift=0, full=True, with_observables=True):
    distance, levels = _lcs_nodes(reference, observed, shift)
    levels = [[node(n) for n in level] for level in levels]
    graph = LCSgraph(reference, observed, shift)
    rec = {"reference": reference, "observed": observed, "shift": shift,
           "edit_distance": edit_distance(reference, observed)}
    g = graph_record(graph)
    assert g["distance"] == distance == rec["edit_distance"]
    if full:
        rec["levels"] = levels
        rec["graph"] = g
        if with_observables:
            rec["observables"] = observables(graph)
        shrink(rec)
    else:
        rec["distance"] = distance
        rec["supremal"] = g["supremal"]
        rec["n_levels"] = len(levels)
        rec["n_nodes"] = len(g["nodes"])
        rec["n_edges"] = len(g["edges"])
        rec["levels_sha256"] = digest(levels)
        rec["graph_sha256"] = digest(g)
    return rec


def sequence_case(reference, observed):
    """from_sequence + extractor outputs."""
    graph = LCSgraph.from_sequence(reference, observed)
    rec = {"reference": reference, "observed": observed, "graph": graph_record(graph),
           "observables": observables(graph), "trim": list(trim(reference, observed))}
    variants = canonical(graph)
    rec["canonical"] = [var(v) for v in variants]
    try:
        rec["hgvs"] = to_hgvs(variants, reference)
    except Exception as exc:  # the reference raises on some of its own vectors
        rec["hgvs_error"] = [type(exc).__name__, str(exc)]
    return shrink(rec)


def shrink(rec, limit=300):
    """Replace bulky parts of a record by their SHA-256 (same canonical JSON as `digest`)."""
    g = rec.get("graph")
    if g is not None and len(g["edges"]) > limit:
        rec["graph_summary"] = {"distance": g["distance"], "supremal": g["supremal"],
                                "n_nodes": len(g["nodes"]), "n_edges": len(g["edges"])}
        rec["graph_sha256"] = digest(g)
        del rec["graph"]
        if "observables" in rec:
            rec["observables_sha256"] = {k: digest(v) for k, v in rec["observables"].items()}
            del rec["observables"]
        if "levels" in rec:
            rec["levels_sha256"] = digest(rec["levels"])
            rec["n_levels"] = len(rec["levels"])
            del rec["levels"]
        if "canonical" in rec and len(rec["canonical"]) > limit:
            rec["canonical_sha256"] = digest(rec["canonical"])
            del rec["canonical"]
    return rec


def params(module, test_name):
    """The argvalues of a @pytest.mark.parametrize on a test function of the reference."""
    fn = getattr(module, test_name)
    for mark in fn.pytestmark:
        if mark.name == "parametrize":
            return list(mark.args[1])
    raise KeyError(test_name)


def main():
    t0 = time.time()
    sys.path.insert(0, os.path.join(REFERENCE, "tests", "relations"))
    from tests.lcs import test_edit_distance as t_ed, test_lcs_graph as t_lg
    from tests.extractor import test_extractor as t_ex, test_local_supremal as t_ls
    from tests.relations import test_sequence_based as t_sb, test_supremal_based as t_sup, test_variant_based as t_vb

    # ---- F1: every vector of the reference's own tests for this path, re-captured by import ----
    pairs = []

    def add(reference, observed, shift=0):
        key = (reference, observed, shift)
        if key not in pairs:
            pairs.append(key)

    for a, b, _ in params(t_ed, "test_edit_distance"):
        add(a, b)
    for a, b, *_ in params(t_lg, "test_lcs_nodes"):
        add(a, b)
    for name in ("test_lcs_graph", "test_lcs_graph_nodes", "test_bfs_traversal", "test_paths"):
        for a, b, _ in params(t_lg, name):
            add(a, b)
    for a, b, *_ in params(t_lg, "test_trim"):
        add(a, b)
    for reference, variants, _ in t_lg.TESTS:
        add(reference, patch(reference, variants))
    extractor_pairs = [(a, b) for a, b, *_ in params(t_ex, "test_extract_sequence")]
    for name in dir(t_ls):
        if name.startswith("test_"):
            try:
                for a, b, *_ in params(t_ls, name):
                    if isinstance(a, str) and isinstance(b, str):
                        extractor_pairs.append((a, b))
            except (KeyError, AttributeError, TypeError):
                pass
    for a, b in extractor_pairs:
        add(a, b)
    seq_triples = [(r, a, b) for r, a, b, _ in t_sb.TESTS]
    for r, a, b in seq_triples:
        add(r, a)
        add(r, b)
        add(a, b)

    ref_cases = [lcs_case(a, b, s) for a, b, s in pairs]
    max_distance = []
    for a, b, md, *_ in params(t_lg, "test_lcs_nodes_max_distance") + params(t_lg, "test_lcs_nodes_max_distance_fail"):
        try:
            d, _ = _lcs_nodes(a, b, max_distance=md)
            max_distance.append({"reference": a, "observed": b, "max_distance": md, "distance": d})
        except ValueError as exc:
            max_distance.append({"reference": a, "observed": b, "max_distance": md, "error": str(exc)})
    from_variant = []
    for reference, variants, expected in t_lg.TESTS:
        g = LCSgraph.from_variant(reference, variants)
        from_variant.append({"reference": reference, "variants": [var(v) for v in variants], "offset": 10,
                             "graph": graph_record(g)})
        assert g.supremal == expected
    for reference, variants, offset, expected in params(t_lg, "test_lcs_graph_from_variant_offset"):
        g = LCSgraph.from_variant(reference, variants, offset)
        from_variant.append({"reference": reference, "variants": [var(v) for v in variants], "offset": offset,
                             "graph": graph_record(g)})
    from_supremal = []
    for reference, _, supremal in t_lg.TESTS:
        g = LCSgraph.from_supremal(reference, supremal)
        from_supremal.append({"reference": reference, "supremal": var(supremal), "graph": graph_record(g)})
    sequences = [sequence_case(a, b) for a, b in dict.fromkeys(extractor_pairs)]
    trims = [{"lhs": a, "rhs": b, "trim": list(trim(a, b))} for a, b, *_ in params(t_lg, "test_trim")]

    relations = {
        "sequence_based": [{"reference": r, "lhs": a, "rhs": b, "relation": sequence_based.compare(r, a, b).value}
                           for r, a, b in seq_triples],
        "variant_based": [{"reference": r, "lhs": [var(v) for v in a], "rhs": [var(v) for v in b],
                           "relation": variant_based.compare(r, a, b).value} for r, a, b, _ in t_vb.TESTS],
        "supremal_based": [{"reference": r, "lhs": var(a), "rhs": var(b),
                            "relation": supremal_based.compare(r, a, b).value} for r, a, b, _ in t_sup.TESTS],
    }

    with open(os.path.join(HERE, "reference_tests.json"), "w") as f:
        json.dump({"lcs": ref_cases, "max_distance": max_distance, "from_variant": from_variant,
                   "from_supremal": from_supremal, "sequences": sequences, "trim": trims, "relations": relations},
                  f, separators=(",", ":"))
    print(f"reference_tests.json: {len(ref_cases)} lcs cases, {len(sequences)} sequence cases  ({time.time()-t0:.1f}s)")

    # ---- F2: seeded random pairs (small: full records; medium: inputs + hashes) ----
    small, medium = [], []
    for case in gen.small_cases():
        small.append(lcs_case(case["reference"], case["observed"], case["shift"], full=True, with_observables=False))
    for case in gen.medium_cases():
        rec = lcs_case(case["reference"], case["observed"], case["shift"], full=False)
        rec["name"] = case["name"]
        medium.append(rec)
    md_cases = []
    for case in gen.max_distance_cases():
        for md in case["limits"]:
            try:
                d, _ = _lcs_nodes(case["reference"], case["observed"], max_distance=md)
                md_cases.append({"reference": case["reference"], "observed": case["observed"], "max_distance": md, "distance": d})
            except ValueError as exc:
                md_cases.append({"reference": case["reference"], "observed": case["observed"], "max_distance": md, "error": str(exc)})
    with open(os.path.join(HERE, "random_small.json"), "w") as f:
        json.dump({"lcs": small, "max_distance": md_cases}, f, separators=(",", ":"))
    with open(os.path.join(HERE, "random_medium.json"), "w") as f:
        json.dump({"lcs": medium}, f, separators=(",", ":"))
    print(f"random_small.json: {len(small)} cases; random_medium.json: {len(medium)} cases  ({time.time()-t0:.1f}s)")

    # ---- sequence-level (from_sequence + canonical + HGVS) on seeded pairs ----
    seqs = []
    for case in gen.sequence_cases():
        rec = sequence_case(case["reference"], case["observed"])
        rec.pop("observables", None)
        rec.pop("observables_sha256", None)
        rec["name"] = case["name"]
        seqs.append(rec)
    rel = []
    for case in gen.relation_cases():
        r, a, b = case["reference"], case["lhs"], case["rhs"]
        rel.append({"reference": r, "lhs": a, "rhs": b, "relation": sequence_based.compare(r, a, b).value})
    # all-vs-all matrix of graph_based.compare over clustered single-variant alleles (C4 in small)
    codes = {relation: code for code, relation in enumerate(Relation)}
    matrices = []
    for params_ in ({}, {"seed": 6007, "length": 250, "n_clusters": 3, "per_cluster": 14, "width": 10}):
        case = gen.matrix_case(**params_)
        graphs = [LCSgraph.from_variant(case["reference"], [Variant(*v)]) for v in case["variants"]]
        labels = [[codes[graph_based.compare(case["reference"], a, b)] for b in graphs] for a in graphs]
        variant_labels = [[codes[variant_based.compare(case["reference"], [Variant(*a)], [Variant(*b)])]
                           for b in case["variants"][:12]] for a in case["variants"][:12]]
        supremal_labels = [[codes[supremal_based.compare(case["reference"], a.supremal, b.supremal)]
                            for b in graphs[:12]] for a in graphs[:12]]
        matrices.append({"params": params_, "reference": case["reference"], "variants": [list(v) for v in case["variants"]],
                         "supremals": [var(g.supremal) for g in graphs], "distances": [g.distance for g in graphs],
                         "labels": labels, "variant_labels_12": variant_labels, "supremal_labels_12": supremal_labels})
    with open(os.path.join(HERE, "random_sequences.json"), "w") as f:
        json.dump({"sequences": seqs, "relations": rel, "matrices": matrices}, f, separators=(",", ":"))
    print(f"random_sequences.json: {len(seqs)} sequence cases, {len(rel)} relation triples  ({time.time()-t0:.1f}s)")

    # ---- F3: large shapes, hashes only; inputs regenerated from the seed by tests/gen.py ----
    large = []
    for case in gen.large_cases():
        t1 = time.time()
        reference, observed = gen.large_pair(case)
        variants, graph = extract_sequence(reference, observed)
        g = graph_record(graph)
        large.append({"case": case, "len_reference": len(reference), "len_observed": len(observed),
                      "input_sha256": digest([reference, observed]),
                      "distance": graph.distance, "supremal_pos": [graph.supremal.start, graph.supremal.end],
                      "n_nodes": len(g["nodes"]), "n_edges": len(g["edges"]),
                      "graph_sha256": digest(g), "canonical_sha256": digest([var(v) for v in variants]),
                      "n_variants": len(variants), "hgvs_sha256": digest(to_hgvs(variants, reference)),
                      "reference_seconds": round(time.time() - t1, 2)})
        print(f"  large {case['name']}: d={graph.distance} nodes={len(g['nodes'])} {time.time()-t1:.1f}s", flush=True)
    with open(os.path.join(HERE, "large.json"), "w") as f:
        json.dump({"from_sequence": large}, f, separators=(",", ":"))
    print(f"large.json: {len(large)} cases  ({time.time()-t0:.1f}s)")


if __name__ == "__main__":
    main()
