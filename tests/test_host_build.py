"""The host half of the graph path without a device: `_build_graph` on caller-provided level lists
(malcs_graphs_from_runs -> malcs_graph_core.hpp, the construction the batched entry points and the device kernel
share) against the golden fixtures captured from the reference and against the oracle's arrays.  The level lists
come from the oracle's restatement of `_lcs_nodes` (pinned in test_oracle.py), off-path runs included."""

import numpy as np
import pytest

from conftest import load_golden
from oracle import oracle as O
from util import assert_graph, graph_record_from_arrays

REF = load_golden("reference_tests.json")
SMALL = load_golden("random_small.json")
MEDIUM = load_golden("random_medium.json")


def _host_graphs(records):
    from algebra_amd import _native
    graphs = [O.lcs_graph(r["reference"], r["observed"], r["shift"]) for r in records]
    m = [len(r["reference"]) for r in records]
    n = [len(r["observed"]) for r in records]
    shift = [r["shift"] for r in records]
    # the oracle lists its runs in the order the sweep meets them: grouped by level (stable) they are the level lists
    runs = [(g.runs - np.array([r["shift"], r["shift"], 0, 0]))[np.argsort(g.runs[:, 3], kind="stable")] if len(g.runs) else g.runs
            for g, r in zip(graphs, records)]
    first = np.concatenate([[0], np.cumsum([len(x) for x in runs])])
    flat = np.concatenate(runs) if runs else np.zeros((0, 4), np.int64)
    n_levels = [g.n_levels if len(g.runs) else 0 for g in graphs]
    return graphs, _native.graphs_from_runs(m, n, shift, n_levels, flat, first)


@pytest.mark.parametrize("name,fixture", [("reference", REF), ("small", SMALL), ("medium", MEDIUM)])
def test_build_graph_from_level_lists(name, fixture):
    records = fixture["lcs"]
    oracle_graphs, res = _host_graphs(records)
    try:
        for i, (rec, ref) in enumerate(zip(records, oracle_graphs)):
            assert tuple(res.supremal[i]) == ref.supremal, (name, i)
            nodes, edges = res.nodes(i), res.edges(i)
            assert np.array_equal(nodes, ref.nodes) and np.array_equal(edges, ref.edges), (name, i)
            if "nodes" in rec or "bfs" in rec:
                assert_graph(rec, graph_record_from_arrays(rec["observed"], ref.distance, tuple(res.supremal[i]), nodes, edges))
    finally:
        res.close()


def test_build_graph_two_part_write():
    """Graphs from 100 000 edges on write their edges in two parts (the nodes' own edges, then the shared prefixes, which
    the worker threads of a batch copy together: write_graph_own / copy_prefixes).  MALCS_SHARED_COPY_EDGES=0 sends every
    fixture graph that way; the library reads the variable once, hence the process of its own."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MALCS_SHARED_COPY_EDGES="0")
    done = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", os.path.abspath(__file__), "-k", "from_level_lists"],
                          cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert done.returncode == 0, done.stdout[-2000:] + done.stderr[-2000:]


def test_build_graph_facade_name():
    # the reference's own use: tests/lcs/test_lcs_graph.py builds graphs from _lcs_nodes output
    from algebra_amd.lcs.lcs_graph import LCSgraph, _build_graph
    for rec in REF["lcs"][:40]:
        a, b, s = rec["reference"], rec["observed"], rec["shift"]
        g = O.lcs_graph(a, b, s)
        levels = [[LCSgraph.Node(*t) for t in lvl] for lvl in g.level_lists()]
        source, supremal = _build_graph(a, b, levels, s)
        lo, hi = g.supremal[2], g.supremal[3]
        assert (supremal.start, supremal.end, supremal.sequence) == (g.supremal[0], g.supremal[1], b[lo:hi])
        assert (source.row, source.col, source.length) == tuple(g.nodes[0])


def test_build_graph_rejects_bad_runs():
    from algebra_amd import _native
    with pytest.raises(_native.MalcsError):
        _native.graphs_from_runs([4], [4], [0], [2], [[0, 0, 5, 0]], [0, 1])      # run longer than the sequences
    with pytest.raises(_native.MalcsError):
        _native.graphs_from_runs([4], [4], [0], [2], [[0, 0, 1, 1], [1, 1, 1, 0]], [0, 2])   # levels descending


def test_sanitizer_run_of_host_graph_code():
    """`make -C oracle asan`: the oracle and the library's graph construction (malcs_graph_core.hpp, arena overflow
    paths included) under AddressSanitizer + UBSan on seeded pairs, compared with each other (tests/checks/asan_core.cpp)."""
    import os
    import shutil
    import subprocess
    if not shutil.which("g++") or not shutil.which("make"):
        pytest.skip("no g++ / make")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    done = subprocess.run(["make", "-C", os.path.join(root, "oracle"), "asan", "ASAN_CASES=400"], capture_output=True, text=True, timeout=600)
    assert done.returncode == 0, done.stdout[-2000:] + done.stderr[-2000:]
    assert "0 mismatches" in done.stdout


def test_thread_sanitizer_run_of_host_threads():
    """`make -C oracle tsan`: the library's kept host threads (malcs_workers) and the packing of symbols into bit planes on
    them under ThreadSanitizer, kept / own / one thread compared word by word (tests/checks/tsan_pack.cpp)."""
    import os
    import shutil
    import subprocess
    if not shutil.which("g++") or not shutil.which("make"):
        pytest.skip("no g++ / make")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    done = subprocess.run(["make", "-C", os.path.join(root, "oracle"), "tsan"], capture_output=True, text=True, timeout=600)
    assert done.returncode == 0, done.stdout[-2000:] + done.stderr[-2000:]
    assert "no findings" in done.stdout
