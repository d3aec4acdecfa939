"""Pins the CPU oracle (oracle/lcs_oracle.c) against the golden fixtures captured from the real
reference: every vector of the reference's own tests for this path plus seeded random pairs."""

import pytest

from conftest import load_golden
from oracle import oracle as O
from util import assert_graph, digest, graph_record_from_arrays
import gen

REF = load_golden("reference_tests.json")
SMALL = load_golden("random_small.json")
MEDIUM = load_golden("random_medium.json")
LARGE = load_golden("large.json")


def oracle_record(reference, observed, shift):
    g = O.lcs_graph(reference, observed, shift)
    assert g.status == 0
    return g, graph_record_from_arrays(observed, g.distance, g.supremal, g.nodes, g.edges)


@pytest.mark.parametrize("idx", range(len(REF["lcs"])))
def test_reference_vectors(idx):
    rec = REF["lcs"][idx]
    a, b, s = rec["reference"], rec["observed"], rec["shift"]
    assert O.edit_distance(a, b) == rec["edit_distance"]
    g, got = oracle_record(a, b, s)
    levels = [[list(t) for t in lvl] for lvl in g.level_lists()]
    if "levels" in rec:
        assert levels == rec["levels"]
    else:
        assert digest(levels) == rec["levels_sha256"]
    assert_graph(rec, got)


def test_random_small():
    for rec in SMALL["lcs"]:
        a, b, s = rec["reference"], rec["observed"], rec["shift"]
        assert O.edit_distance(a, b) == rec["edit_distance"]
        assert O.edit_distance(b, a) == rec["edit_distance"]
        g, got = oracle_record(a, b, s)
        levels = [[list(t) for t in lvl] for lvl in g.level_lists()]
        if "levels" in rec:
            assert levels == rec["levels"], (a, b, s)
        else:
            assert digest(levels) == rec["levels_sha256"]
        assert_graph(rec, got)


@pytest.mark.parametrize("idx", range(len(MEDIUM["lcs"])))
def test_random_medium(idx):
    rec = MEDIUM["lcs"][idx]
    a, b, s = rec["reference"], rec["observed"], rec["shift"]
    assert O.edit_distance(a, b) == rec["edit_distance"]
    g, got = oracle_record(a, b, s)
    assert g.distance == rec["distance"]
    assert got["supremal"] == rec["supremal"]
    assert g.n_levels == rec["n_levels"]
    assert digest([[list(t) for t in lvl] for lvl in g.level_lists()]) == rec["levels_sha256"]
    assert digest(got) == rec["graph_sha256"]


def test_max_distance():
    for rec in REF["max_distance"] + SMALL["max_distance"]:
        g = O.lcs_graph(rec["reference"], rec["observed"], 0, rec["max_distance"])
        if "error" in rec:
            assert g.status == 1 and rec["error"] == "maximum distance exceeded"
        else:
            assert g.status == 0 and g.distance == rec["distance"]


def test_trim():
    for rec in REF["trim"]:
        assert list(O.trim(rec["lhs"], rec["rhs"])) == rec["trim"]
    for rec in REF["sequences"]:
        assert list(O.trim(rec["reference"], rec["observed"])) == rec["trim"]


def test_from_supremal_vectors():
    for rec in REF["from_supremal"]:
        start, end, seq = rec["supremal"]
        _, got = oracle_record(rec["reference"][start:end], seq, start)
        assert got == rec["graph"]


def test_optimal_runs_in_sweep_order_rebuild_the_graph():
    """dropping off-path runs keeps the graph; this is what the device relies on."""
    import numpy as np
    for rec in SMALL["lcs"][:200]:
        a, b, s = rec["reference"], rec["observed"], rec["shift"]
        g = O.lcs_graph(a, b, s)
        if g.n_levels == 0 or len(g.runs) == 0:
            continue
        on_path = {tuple(n) for n in g.nodes.tolist()}
        # keep only runs that overlap a node of the final graph (cheap superset of the optimal runs)
        keep = []
        for row, col, length, level in g.runs.tolist():
            for r2, c2, l2 in on_path:
                if col - row == c2 - r2 and r2 >= row and r2 <= row + length:
                    keep.append((row, col, length, level))
                    break
        if not keep or max(k[3] for k in keep) != g.n_levels - 1:
            continue
        g2 = O.graph_from_runs(len(a), len(b), s, np.array(keep), g.n_levels, g.distance)
        assert g2.supremal == g.supremal and g2.nodes.tolist() == g.nodes.tolist() and g2.edges.tolist() == g.edges.tolist()


@pytest.mark.parametrize("idx", range(len(LARGE["from_sequence"])))
def test_large_inputs_regenerate(idx):
    rec = LARGE["from_sequence"][idx]
    a, b = gen.large_pair(rec["case"])
    assert digest([a, b]) == rec["input_sha256"]
    assert O.edit_distance(a, b) == rec["distance"]


def test_python_restatement_of_edit_distance():
    from oracle import py_oracle as P
    for rec in REF["lcs"] + SMALL["lcs"][:200] + MEDIUM["lcs"][:20]:
        assert P.edit_distance(rec["reference"], rec["observed"]) == rec["edit_distance"]
