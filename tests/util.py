"""Shared helpers of the test-suite: canonical records and their digests.

`graph_record*` produce exactly the structure tests/golden/make_golden.py captured from the
reference, so records (or their SHA-256) can be compared directly.
"""

import hashlib
import json


def digest(obj):
    return hashlib.sha256(json.dumps(obj, separators=(",", ":"), sort_keys=True).encode()).hexdigest()


def graph_record_from_arrays(observed, distance, supremal, nodes, edges):
    """Arrays as returned by the oracle / libmalcs -> canonical record."""
    s, e, lo, hi = (int(x) for x in supremal)
    return {
        "distance": int(distance),
        "supremal": [s, e, observed[lo:hi]],
        "nodes": [[int(v) for v in row] for row in nodes.tolist()],
        "edges": [[int(a), int(b), int(c), int(d), observed[int(lo_):int(hi_)]] for a, b, c, d, lo_, hi_ in edges.tolist()],
    }


def graph_record(graph):
    """LCSgraph-like object (facade) -> canonical record: BFS by identity following edge order."""
    ids = {id(graph._source): 0}
    order = [graph._source]
    edges = []
    head = 0
    while head < len(order):
        src = order[head]
        for child, variant in src.edges:
            if id(child) not in ids:
                ids[id(child)] = len(order)
                order.append(child)
            edges.append([head, ids[id(child)], variant.start, variant.end, variant.sequence])
        head += 1
    return {
        "distance": graph.distance,
        "supremal": [graph.supremal.start, graph.supremal.end, graph.supremal.sequence],
        "nodes": [[n.row, n.col, n.length] for n in order],
        "edges": edges,
    }


def observables(graph):
    def node(n):
        return [n.row, n.col, n.length]

    def var(v):
        return [v.start, v.end, v.sequence]

    return {
        "nodes_order": [node(n) for n in graph.nodes()],
        "bfs": [[node(s), node(t)] + var(v[0]) for s, t, v in graph.bfs_traversal()],
        "edges_sorted": sorted(var(v) for v in graph.edges()),
        "uniq_atomics": sorted([i, ch] for i, ch in graph.uniq_atomics()),
    }


def expected_graph(rec):
    """(record or None, sha256 or None) from a fixture entry in either full or shrunk form."""
    if "graph" in rec:
        return rec["graph"], None
    return None, rec["graph_sha256"]


def assert_graph(rec, got):
    want, sha = expected_graph(rec)
    if want is not None:
        assert got == want
    else:
        assert digest(got) == sha
