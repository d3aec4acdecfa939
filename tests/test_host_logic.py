"""CPU tests of the host-side logic above the C-ABI: extractor (canonical, HGVS), relation
arithmetic and the tiled/sharded relation matrix.  Graph arrays come from the golden fixtures or
from the CPU oracle (allowed in tests); distances for the matrix come from the oracle through the
`distance_fn` hook.  The GPU versions of the same checks live in test_gpu_relations.py."""

import time

import numpy as np
import pytest
from util import HostTiles

from conftest import load_golden
from oracle import oracle as O

REF = load_golden("reference_tests.json")
SEQS = load_golden("random_sequences.json")


def facade_graph_from_record(observed_unused, rec):
    """Rebuild a facade LCSgraph from a canonical fixture record."""
    from algebra_amd.lcs import LCSgraph
    from algebra_amd.variants import Variant
    self = LCSgraph.__new__(LCSgraph)
    self.distance = rec["distance"]
    self.supremal = Variant(*rec["supremal"])
    self._result = None
    made = [LCSgraph.Node(*n) for n in rec["nodes"]]
    for src, dst, start, end, seq in rec["edges"]:
        made[src].edges.append((made[dst], Variant(start, end, seq)))
    self._source = made[0]
    return self


def oracle_graph(reference, observed, shift=0):
    from algebra_amd.lcs import LCSgraph
    g = O.lcs_graph(reference, observed, shift)
    return LCSgraph._from_arrays(observed, g.distance, g.supremal, nodes=g.nodes, edges=g.edges)


def test_canonical_and_hgvs_on_fixture_graphs():
    from algebra_amd.extractor import canonical, to_hgvs
    checked = 0
    for rec in REF["sequences"] + SEQS["sequences"]:
        if "graph" not in rec:
            continue
        graph = facade_graph_from_record(rec["observed"], rec["graph"])
        variants = canonical(graph)
        if "canonical" in rec:
            assert [[v.start, v.end, v.sequence] for v in variants] == rec["canonical"], (rec["reference"], rec["observed"])
        if "hgvs" in rec:
            assert to_hgvs(variants, rec["reference"]) == rec["hgvs"]
        else:
            with pytest.raises(Exception) as exc:
                to_hgvs(variants, rec["reference"])
            assert [type(exc.value).__name__, str(exc.value)] == rec["hgvs_error"]
        checked += 1
    assert checked > 150


def test_traversals_on_fixture_graphs():
    from util import observables
    for rec in REF["lcs"]:
        if "graph" in rec and "observables" in rec:
            graph = facade_graph_from_record(rec["observed"], rec["graph"])
            assert observables(graph) == rec["observables"]


def test_relation_arithmetic():
    from algebra_amd.relations.graph_based import from_distances, union_windows
    from algebra_amd.relations import Relation
    from algebra_amd.variants import Variant
    assert from_distances(2, 3, 5) == Relation.DISJOINT
    assert from_distances(5, 2, 3) == Relation.CONTAINS
    assert from_distances(2, 5, 3) == Relation.IS_CONTAINED
    assert from_distances(2, 2, 2) is None
    assert from_distances(0, 0, 0) == Relation.DISJOINT          # order of the tests matters (reference :67-74)
    assert union_windows("ACGTACGT", Variant(1, 2, "TT"), Variant(4, 6, "")) == ("TTGTAC", "CGT")
    assert [r.value for r in Relation] == ["equivalent", "contains", "is_contained", "overlap", "disjoint"]


def matrix_inputs(case):
    from algebra_amd.variants import Variant
    graphs = []
    for (start, end, seq) in case["supremals"]:
        graphs.append(oracle_graph(case["reference"][start:end], seq, start))
    assert [g.distance for g in graphs] == case["distances"]
    assert [[g.supremal.start, g.supremal.end, g.supremal.sequence] for g in graphs] == case["supremals"]
    return graphs


def oracle_distances(pairs):
    return [O.edit_distance(a, b) for a, b in pairs]


@pytest.mark.parametrize("tile", [7, 16, 256])
def test_compare_matrix_single_rank(tile):
    from algebra_amd.relations import compare_matrix
    for case in SEQS["matrices"]:
        stats = {}
        labels = compare_matrix(case["reference"], matrix_inputs(case), tile=tile, distance_fn=oracle_distances, stats=stats, tiles_cls=HostTiles)
        assert labels.tolist() == case["labels"]
        assert 0 < stats["dp_pairs"] < len(case["labels"]) ** 2 // 2


def _rank_main(rank, world, port, out):
    import os
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from algebra_amd.relations import compare_matrix
        case = SEQS["matrices"][0]
        stats = {}
        from util import gloo_gather
        labels = compare_matrix(case["reference"], matrix_inputs(case), tile=8, rank=rank, world=world, gather=gloo_gather(dist),
                                distance_fn=oracle_distances, stats=stats, tiles_cls=HostTiles)
        out.put((rank, labels.tolist() == case["labels"], stats["tiles"]))
    finally:
        dist.destroy_process_group()


def test_compare_matrix_two_ranks_gloo():
    """The N > 1 path: tiles sharded over 2 ranks, one all-gather of uint8 tiles (gloo on CPU)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = 29500 + (np.random.RandomState().randint(0, 2000))
    procs = [ctx.Process(target=_rank_main, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    results = sorted(out.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[1] for r in results] == [True, True]
    n_tiles = len(__import__("algebra_amd.relations.matrix", fromlist=["upper_tiles"]).upper_tiles(len(SEQS["matrices"][0]["labels"]), 8))
    assert results[0][2] + results[1][2] == n_tiles and abs(results[0][2] - results[1][2]) <= 1


def test_shard_partition():
    from algebra_amd.relations.matrix import shard, upper_tiles
    tiles = upper_tiles(1000, 256)
    assert len(tiles) == 10
    for world in (1, 2, 3, 8):
        parts = [shard(tiles, r, world) for r in range(world)]
        assert sorted(t for p in parts for t in p) == sorted(tiles)


def test_graph_batch_keeps_columns_only_while_it_holds_the_same_graphs():
    """GraphBatch (what the batch constructors return) is a list that keeps the per-variant columns of its matrix: made
    once, found again for the same graphs and reference, made anew after any change of either."""
    from algebra_amd.lcs.lcs_graph import GraphBatch
    made = []

    def make(reference, graphs):
        made.append((reference, tuple(graphs)))
        return len(made)

    items = [object() for _ in range(5)]
    batch = GraphBatch(items)
    assert isinstance(batch, list) and batch == items
    assert batch.columns("ACGT", make) == 1
    assert batch.columns("ACGT", make) == 1 and len(made) == 1
    assert batch.columns("ACGA", make) == 2                      # another reference
    assert batch.columns("ACGA", make) == 2
    batch[1], batch[2] = batch[2], batch[1]                      # same graphs, another order
    assert batch.columns("ACGA", make) == 3
    batch.append(object())
    assert batch.columns("ACGA", make) == 4
    batch.pop()
    assert batch.columns("ACGA", make) == 5                      # what was kept belonged to six graphs
    assert batch.columns("ACGA", make) == 5
    assert type(batch[1:3]) is list and type(batch + batch) is list and not hasattr(batch[1:3], "columns")
    assert GraphBatch().columns("", make) == 6


def test_communicator_id_handshake_turns_strangers_away():
    """algebra_amd.comm: rank 0 hands the communicator's id only to connections that say which rank of which job they are;
    a stray connection and a rank of another job do not use up a place (they used to, and a real rank then waited for
    its time limit)."""
    import socket
    import threading
    from algebra_amd import comm
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    uid = bytes(range(128))
    got, errors = {}, []

    def serve():
        try:
            comm._serve(uid, "127.0.0.1", port, 3, 20, "job-a")
        except Exception as e:                                  # noqa: BLE001
            errors.append(e)

    server = threading.Thread(target=serve)
    server.start()
    # a stranger that says nothing, and a rank of another job: both are turned away
    deadline = time.time() + 10
    while True:
        try:
            stray = socket.create_connection(("127.0.0.1", port), timeout=5)
            break
        except OSError:
            assert time.time() < deadline
            time.sleep(0.05)
    stray.close()
    with socket.create_connection(("127.0.0.1", port), timeout=5) as other:
        other.sendall(comm._hello(1, 3, "job-b"))
        other.settimeout(5)
        assert other.recv(128) == b""
    for rank in (2, 1):
        got[rank] = comm._fetch("127.0.0.1", port, 10, rank, 3, "job-a")
    server.join(20)
    assert not server.is_alive() and not errors
    assert got == {1: uid, 2: uid}
