"""GPU parity tests of the callers of the LCS path: relations (four flavours, matrix) and the
extractor, against fixtures captured from the reference."""

import numpy as np
import pytest

from conftest import load_golden
from util import digest
import gen

pytestmark = pytest.mark.gpu

REF = load_golden("reference_tests.json")
SEQS = load_golden("random_sequences.json")
LARGE = load_golden("large.json")


def test_sequence_based_relations():
    from algebra_amd.relations import Relation, sequence_based as sb
    for rec in REF["relations"]["sequence_based"] + SEQS["relations"]:
        r, a, b = rec["reference"], rec["lhs"], rec["rhs"]
        want = Relation(rec["relation"])
        assert sb.compare(r, a, b) == want, (r, a, b)
    for rec in REF["relations"]["sequence_based"]:
        r, a, b = rec["reference"], rec["lhs"], rec["rhs"]
        want = Relation(rec["relation"])
        assert sb.are_equivalent(r, a, b) == (want == Relation.EQUIVALENT)
        assert sb.contains(r, a, b) == (want == Relation.CONTAINS)
        assert sb.is_contained(r, a, b) == (want == Relation.IS_CONTAINED)
        assert sb.are_disjoint(r, a, b) == (want == Relation.DISJOINT)
        assert sb.have_overlap(r, a, b) == (want == Relation.OVERLAP)


def test_variant_and_supremal_based_relations():
    import algebra_amd
    from algebra_amd.relations import Relation, supremal_based, variant_based
    from algebra_amd.variants import Variant
    for rec in REF["relations"]["variant_based"]:
        lhs, rhs = [Variant(*v) for v in rec["lhs"]], [Variant(*v) for v in rec["rhs"]]
        want = Relation(rec["relation"])
        assert algebra_amd.compare(rec["reference"], lhs, rhs) == want
        assert variant_based.contains(rec["reference"], lhs, rhs) == (want == Relation.CONTAINS)
        assert variant_based.are_disjoint(rec["reference"], lhs, rhs) == (want == Relation.DISJOINT)
    for rec in REF["relations"]["supremal_based"]:
        lhs, rhs = Variant(*rec["lhs"]), Variant(*rec["rhs"])
        want = Relation(rec["relation"])
        assert supremal_based.compare(rec["reference"], lhs, rhs) == want
        assert supremal_based.are_equivalent(rec["reference"], lhs, rhs) == (want == Relation.EQUIVALENT)
        assert supremal_based.contains(rec["reference"], lhs, rhs) == (want == Relation.CONTAINS)
        assert supremal_based.is_contained(rec["reference"], lhs, rhs) == (want == Relation.IS_CONTAINED)
        assert supremal_based.are_disjoint(rec["reference"], lhs, rhs) == (want == Relation.DISJOINT)
        assert supremal_based.have_overlap(rec["reference"], lhs, rhs) == (want == Relation.OVERLAP)
    codes = list(Relation)
    case = SEQS["matrices"][0]
    alleles = [[Variant(*v)] for v in case["variants"][:12]]
    sup = [Variant(*v) for v in case["supremals"][:12]]
    for i in range(12):
        for j in range(12):
            assert variant_based.compare(case["reference"], alleles[i], alleles[j]) == codes[case["variant_labels_12"][i][j]]
            assert supremal_based.compare(case["reference"], sup[i], sup[j]) == codes[case["supremal_labels_12"][i][j]]


@pytest.mark.parametrize("n,tile,world", [(1, 4, 1), (37, 8, 1), (300, 16, 3), (1000, 256, 2), (513, 64, 8), (700, 7, 5)])
def test_relation_tile_kernels(n, tile, world):
    """Kernel family 3 through the C-ABI against the numpy restatement: pre-filter labels and undecided pairs of
    every rank's tiles, scatter of verdicts, assembly of the full matrix from the gathered blocks."""
    from algebra_amd import _native
    from util import HostTiles
    rs = np.random.RandomState(n + tile)
    start = rs.randint(0, 4 * n, n).astype(np.int64)
    end = start + rs.randint(0, 12, n)
    symmask = rs.randint(0, 16, n).astype(np.uint64) | (rs.randint(0, 2, n).astype(np.uint64) << np.uint64(63))
    key = np.arange(n, dtype=np.uint32)
    twins = rs.randint(0, n, max(1, n // 10))
    key[twins] = key[(twins * 7) % n]; start[twins] = start[(twins * 7) % n]; end[twins] = end[(twins * 7) % n]
    symmask[twins] = symmask[(twins * 7) % n]
    ctx = _native.default_context()
    dev = [_native.RelationMatrix(ctx, start, end, key, symmask, tile, r, world) for r in range(world)]
    host = [HostTiles(start, end, key, symmask, tile, r, world) for r in range(world)]
    blocks_d, blocks_h = [], []
    for d, h in zip(dev, host):
        assert np.array_equal(d.tiles(), h.tiles())
        pd, ph = d.pending(), h.pending()
        assert sorted(map(tuple, pd.tolist())) == sorted(map(tuple, ph.tolist()))
        assert (pd[:, 0] < pd[:, 1]).all()
        codes = {(i, j): int((i * 31 + j) % 5) for i, j in ph.tolist()}
        d.resolve(np.array([codes[(i, j)] for i, j in pd.tolist()], np.uint8))
        h.resolve(np.array([codes[(i, j)] for i, j in ph.tolist()], np.uint8))
        assert np.array_equal(d.tiles(), h.tiles())
        blocks_d.append(d.tiles()); blocks_h.append(h.tiles())
    gathered = np.concatenate(blocks_d) if world > 1 else None
    want = host[0].assemble(np.concatenate(blocks_h) if world > 1 else None)
    for d in dev:
        assert np.array_equal(d.assemble(gathered), want)
        d.close()
    if n > 1:
        with pytest.raises(_native.MalcsError):
            bad = _native.RelationMatrix(ctx, start, end, key, symmask, tile, 0, 1)
            try:
                bad.resolve(np.full(len(bad.pending()) or 1, 9, np.uint8)) if len(bad.pending()) else (_ for _ in ()).throw(_native.MalcsError("none"))
            finally:
                bad.close()


@pytest.mark.parametrize("tile", [16, 256])
def test_compare_matrix(tile):
    """C4 in small: supremal graphs on the GPU, then the tiled all-vs-all matrix."""
    from algebra_amd.lcs import LCSgraph
    from algebra_amd.relations import compare_matrix
    from algebra_amd.variants import Variant
    for case in SEQS["matrices"]:
        graphs = [LCSgraph.from_variant(case["reference"], [Variant(*v)]) for v in case["variants"]]
        assert [[g.supremal.start, g.supremal.end, g.supremal.sequence] for g in graphs] == case["supremals"]
        assert [g.distance for g in graphs] == case["distances"]
        stats = {}
        labels = compare_matrix(case["reference"], graphs, tile=tile, stats=stats)
        assert labels.tolist() == case["labels"]
        assert stats["dp_pairs"] > 0


def test_compare_matrix_kept_columns():
    """A batch of graphs keeps the per-variant columns of its matrix (GraphBatch.columns): the second matrix over the
    same batch re-uses them and gives the same labels, a batch that was changed in between makes them anew, and a
    sub-list (a plain list) never sees them.  The matrix itself sits in pinned host memory and is an ordinary array."""
    from algebra_amd.lcs import LCSgraph
    from collections.abc import MutableSequence
    from algebra_amd.lcs.lcs_graph import VariantGraphBatch
    from algebra_amd.relations import compare_matrix
    from algebra_amd.variants import Variant
    case = SEQS["matrices"][0]
    want = np.array(case["labels"], dtype=np.uint8)
    graphs = LCSgraph.from_variant_batch(case["reference"], [[Variant(*v)] for v in case["variants"]])
    assert isinstance(graphs, VariantGraphBatch) and isinstance(graphs, MutableSequence)   # the library ran the widening rounds
    assert not any(graphs._made)                                 # ... and no graph object exists yet
    first = compare_matrix(case["reference"], graphs, tile=16)
    columns = graphs._columns[2]
    second = compare_matrix(case["reference"], graphs, tile=16)
    assert graphs._columns[2] is columns
    assert np.array_equal(first, want) and np.array_equal(second, want)
    first[0, 0] = 7                                              # writable, and not shared with the next result
    assert compare_matrix(case["reference"], graphs, tile=16)[0, 0] == 0
    # the batch changes: two graphs swapped, then one dropped
    assert not any(graphs._made)                                 # three matrices later: still arrays only
    graphs[0], graphs[5] = graphs[5], graphs[0]
    order = [5] + list(range(1, 5)) + [0] + list(range(6, len(graphs)))
    assert np.array_equal(compare_matrix(case["reference"], graphs, tile=16), want[np.ix_(order, order)])
    assert graphs._list._columns[2] is not columns
    del graphs[3]
    order.pop(3)
    assert np.array_equal(compare_matrix(case["reference"], graphs, tile=16), want[np.ix_(order, order)])
    part = graphs[2:9]
    assert type(part) is list
    assert np.array_equal(compare_matrix(case["reference"], part, tile=16), want[np.ix_(order[2:9], order[2:9])])
    # another reference of the same length is another matrix: the columns kept for the first must not be used
    other = case["reference"][::-1]
    graphs2 = LCSgraph.from_variant_batch(other, [[Variant(*v)] for v in case["variants"]])
    plain = compare_matrix(other, list(graphs2), tile=16)
    compare_matrix(other, graphs2, tile=16)
    assert np.array_equal(compare_matrix(other, graphs2, tile=16), plain)


def test_from_variant_batch_equals_singles():
    from algebra_amd.lcs import LCSgraph
    from algebra_amd.variants import Variant
    from util import graph_record
    case = SEQS["matrices"][1]
    alleles = [[Variant(*v)] for v in case["variants"]] + [[], [Variant(3, 4, case["reference"][3])]]
    alleles.append([Variant(5, 6, "T" if case["reference"][5] != "T" else "G"), Variant(40, 42, "")])
    batch = LCSgraph.from_variant_batch(case["reference"], alleles, offset=2)
    for allele, got in zip(alleles, batch):
        assert graph_record(got) == graph_record(LCSgraph.from_variant(case["reference"], allele, offset=2))
    assert [[g.supremal.start, g.supremal.end, g.supremal.sequence] for g in batch[:len(case["supremals"])]] == case["supremals"]
    for rec in REF["from_variant"]:
        got = LCSgraph.from_variant_batch(rec["reference"], [[Variant(*v) for v in rec["variants"]]], rec["offset"])[0]
        assert graph_record(got) == rec["graph"]


def test_from_variant_batch_other_alphabets_and_fallback():
    """The library's widening rounds (malcs_from_variant_batch) over symbols that are not A, C, G, T -- its own code
    table, columns without atom masks -- and the facade's own rounds where the library's entry does not apply (a symbol
    beyond Latin-1): the same graphs as from_variant one by one, the same matrix as compare() pair by pair."""
    from algebra_amd.lcs import LCSgraph
    from algebra_amd.lcs.lcs_graph import GraphBatch, VariantGraphBatch
    from algebra_amd.relations import compare_matrix
    from algebra_amd.relations.graph_based import compare
    from algebra_amd.relations.relation import CODES
    from algebra_amd.variants import Variant
    from util import graph_record
    rs = np.random.RandomState(77)
    for alphabet, kind in (("ACDEFGHIKLMNPQRSTVWY", VariantGraphBatch), ("AC\u0416T", GraphBatch)):
        reference = "".join(alphabet[k] for k in rs.randint(0, len(alphabet), 300))
        alleles = []
        for _ in range(40):
            start = int(rs.randint(0, 290))
            end = start + int(rs.randint(0, 6))
            ins = "".join(alphabet[k] for k in rs.randint(0, len(alphabet), rs.randint(0, 5)))
            alleles.append([Variant(start, end, ins)])
        alleles += [[], [Variant(10, 12, reference[10:12])], [Variant(0, 0, alphabet[1]), Variant(299, 300, "")]]
        batch = LCSgraph.from_variant_batch(reference, alleles, offset=3)
        assert type(batch) is kind and len(batch) == len(alleles)
        singles = [LCSgraph.from_variant(reference, allele, offset=3) for allele in alleles]
        for got, want in zip(batch, singles):
            assert graph_record(got) == graph_record(want)
        labels = compare_matrix(reference, batch, tile=16)
        for i in range(0, len(alleles), 5):
            for j in range(len(alleles)):
                assert labels[i, j] == CODES[compare(reference, singles[i], singles[j])], (i, j)


def test_from_variant_batch_bad_arguments():
    from algebra_amd import _native
    ctx = _native.default_context()
    ref = np.zeros(20, np.uint8)
    with pytest.raises(_native.MalcsError, match="not an interval of the reference"):
        _native.VariantGraphs(ctx, ref, [5], [30], np.zeros(1, np.uint8), [0, 0], 10, 4)
    with pytest.raises(_native.MalcsError, match="not an interval of the reference"):
        _native.VariantGraphs(ctx, ref, [7], [5], np.zeros(1, np.uint8), [0, 0], 10, 4)
    with pytest.raises(_native.MalcsError, match="n_symbols"):
        _native.VariantGraphs(ctx, ref, [5], [6], np.zeros(1, np.uint8), [0, 1], 10, 0)
    none = _native.VariantGraphs(ctx, ref, [], [], np.zeros(1, np.uint8), [0], 10, 4)
    assert none.n == 0 and none.n_rounds == 0
    key, symmask, folded, sup_off, sup, atom_off, atoms = none.matrix_columns(with_atoms=True)
    assert len(key) == 0 and sup_off.tolist() == [0] and not folded
    none.close()


def test_extract_sequence_batch_equals_singles():
    """`extract_sequence_batch` / `LCSgraph.from_sequence_batch`: every reference-test pair in shared batches."""
    from algebra_amd.extractor import extract_sequence_batch, to_hgvs
    from util import graph_record
    records = REF["sequences"] + SEQS["sequences"][:60]
    got = extract_sequence_batch([(rec["reference"], rec["observed"]) for rec in records])
    from util import assert_graph
    for rec, (variants, graph) in zip(records, got):
        assert_graph(rec, graph_record(graph))
        if "canonical" in rec:
            assert [[v.start, v.end, v.sequence] for v in variants] == rec["canonical"]
        if "hgvs" in rec:
            assert to_hgvs(variants, rec["reference"]) == rec["hgvs"]


def test_extract_sequence_canonical_and_hgvs():
    from algebra_amd.extractor import extract_sequence, to_hgvs
    for rec in REF["sequences"] + SEQS["sequences"]:
        variants, graph = extract_sequence(rec["reference"], rec["observed"])
        got = [[v.start, v.end, v.sequence] for v in variants]
        if "canonical" in rec:
            assert got == rec["canonical"], (rec["reference"], rec["observed"])
        else:
            assert digest(got) == rec["canonical_sha256"]
        if "hgvs" in rec:
            assert to_hgvs(variants, rec["reference"]) == rec["hgvs"]
        else:
            with pytest.raises(Exception) as exc:
                to_hgvs(variants, rec["reference"])
            assert [type(exc.value).__name__, str(exc.value)] == rec["hgvs_error"]


def test_library_canonical_equals_python_canonical():
    """malcs_graphs_canonical (C++ on arrays) against the Python body of extractor.canonical on the same graphs."""
    from algebra_amd.extractor import canonical
    from algebra_amd.lcs import LCSgraph
    recs = REF["sequences"] + SEQS["sequences"]
    graphs = LCSgraph.batch([(r["reference"], r["observed"]) for r in recs])
    for rec, graph in zip(recs, graphs):
        fast = canonical(graph)
        graph._source                      # materialise, then hide the handle: forces the Python body
        graph._result = None
        assert canonical(graph) == fast, (rec["reference"], rec["observed"])


def test_extract_like_the_reference_tests():
    from algebra_amd.extractor import extract, extract_sequence, extract_supremal, to_hgvs
    from algebra_amd.variants import Variant
    rec = max((r for r in REF["sequences"] if "hgvs" in r), key=lambda r: len(r["reference"]))
    variants, _ = extract_sequence(rec["reference"], rec["observed"])
    assert to_hgvs(variants, rec["reference"]) == rec["hgvs"]
    variants, graph = extract("GTGTGTTTTTTTAACAGGGA", [Variant(8, 9, "")])
    assert graph.supremal == Variant(5, 12, "TTTTTT")
    variants2, _ = extract_supremal("GTGTGTTTTTTTAACAGGGA", graph.supremal)
    assert variants2 == variants


@pytest.mark.parametrize("idx", range(len(LARGE["from_sequence"])))
def test_extract_sequence_large(idx):
    """BASELINE configs 3 and 5 (i): extract_sequence on 10 kb tandem-repeat windows and on 100 kb / 1 Mb pairs."""
    from algebra_amd.extractor import extract_sequence, to_hgvs
    rec = LARGE["from_sequence"][idx]
    a, b = gen.large_pair(rec["case"])
    variants, graph = extract_sequence(a, b)
    assert graph.distance == rec["distance"]
    assert len(variants) == rec["n_variants"]
    assert digest([[v.start, v.end, v.sequence] for v in variants]) == rec["canonical_sha256"]
    assert digest(to_hgvs(variants, a)) == rec["hgvs_sha256"]


VARIANTS = load_golden("variants.json")


def test_to_dot_matches_the_reference():
    """Graphviz export (reference algebra/utils.py:20-62) over graphs built on the GPU, all label flavours."""
    from algebra_amd.lcs import LCSgraph
    from algebra_amd.utils import to_dot
    from algebra_amd.variants import Variant
    for rec in VARIANTS["to_dot"]:
        if rec["variants"] is None:
            graph = LCSgraph("", "")
        else:
            graph = LCSgraph.from_variant(rec["reference"], [Variant(*v) for v in rec["variants"]])
        got = list(to_dot(rec["reference"], graph, labels=rec["labels"], hgvs=rec["hgvs"], atomics=rec["atomics"]))
        assert got == rec["dot"], (rec["reference"], rec["variants"])


def test_config1_plumbing():
    """BASELINE config 1 in small: HGVS text -> parse_hgvs -> compare -> Relation, labels captured from the reference."""
    from algebra_amd import compare
    from algebra_amd.variants import parse_hgvs
    case = VARIANTS["config1"]
    reference = case["reference"]
    for rec in case["pairs"] + case["hot_pairs"]:
        lhs, rhs = parse_hgvs(rec["lhs_hgvs"], reference), parse_hgvs(rec["rhs_hgvs"], reference)
        assert [[v.start, v.end, v.sequence] for v in lhs] == rec["lhs"]
        assert compare(reference, lhs, rhs).value == rec["relation"], (rec["lhs_hgvs"], rec["rhs_hgvs"])


def test_config1_batched():
    """The same pairs through compare_batch (shared widening rounds + one distance batch)."""
    from algebra_amd.relations import compare_batch
    from algebra_amd.variants import parse_hgvs
    case = VARIANTS["config1"]
    reference = case["reference"]
    recs = case["pairs"] + case["hot_pairs"]
    lhs = [parse_hgvs(r["lhs_hgvs"], reference) for r in recs]
    rhs = [parse_hgvs(r["rhs_hgvs"], reference) for r in recs]
    got = compare_batch(reference, lhs, rhs)
    assert [g.value for g in got] == [r["relation"] for r in recs]


def test_atom_masks_equal_atom_sets():
    """LCSgraph.is_disjoint over per-position bit masks (edge arrays) against the set-based definition
    (reference lcs_graph.py:207-228), on clustered variants where all three outcomes occur."""
    from algebra_amd.lcs import LCSgraph
    from algebra_amd.variants import Variant
    bits = {"": 0, "A": 1, "C": 2, "G": 3, "T": 4}
    seen = set()
    for case in SEQS["matrices"]:
        graphs = LCSgraph.from_variant_batch(case["reference"], [[Variant(*v)] for v in case["variants"]])
        for g in graphs:
            lo, masks = g._atom_masks()
            want = {(idx, ch) for idx, ch in g.uniq_atomics()}
            got = {(lo + int(k), ch) for k in range(len(masks)) for ch, b in bits.items() if masks[k] >> b & 1}
            assert got == want
        for i in range(0, len(graphs), 3):
            for j in range(len(graphs)):
                want = graphs[i].uniq_atomics().isdisjoint(graphs[j].uniq_atomics())
                assert graphs[i].is_disjoint(graphs[j]) == want
                seen.add(want)
    assert seen == {True, False}
    # larger graphs: tandem repeats with thousands of edges
    rs = np.random.RandomState(77)
    a, b = gen.tandem_pair(rs, 600, 4)
    c = a[:300] + "ACGT" + a[300:]
    g1, g2 = LCSgraph.batch([(a, b), (a, c)])
    assert g1.is_disjoint(g2) == g1.uniq_atomics().isdisjoint(g2.uniq_atomics())


def _matrix_rank(rank, world, port, out):
    import os
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), MALCS_DEVICE="0")
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from algebra_amd.lcs import LCSgraph
        from algebra_amd.relations import compare_matrix
        from algebra_amd.variants import Variant
        case = SEQS["matrices"][0]
        graphs = LCSgraph.from_variant_batch(case["reference"], [[Variant(*v)] for v in case["variants"]])
        stats = {}
        from util import gloo_gather
        labels = compare_matrix(case["reference"], graphs, tile=8, rank=rank, world=world, gather=gloo_gather(dist), stats=stats)
        out.put((rank, labels.tolist() == case["labels"], stats["tiles"], stats["dp_pairs"]))
    finally:
        dist.destroy_process_group()


def test_compare_matrix_two_ranks_on_the_device():
    """The N > 1 product path end to end: two ranks (both on GPU 0, gloo for the rendezvous since one card cannot
    host two RCCL ranks), each with its own tiles on the device, all-gather of the label blocks, device assembly."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = 31500 + np.random.RandomState().randint(0, 2000)
    procs = [ctx.Process(target=_matrix_rank, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    results = sorted(out.get(timeout=150) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[1] for r in results] == [True, True]
    assert abs(results[0][2] - results[1][2]) <= 1 and results[0][3] + results[1][3] > 0


def _matrix_rccl_rank(port, out):
    import os
    import sys
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), MALCS_DEVICE="0", RANK="0", WORLD_SIZE="1")
    from algebra_amd import comm as job
    from algebra_amd.lcs import LCSgraph
    from algebra_amd.relations import compare_matrix
    from algebra_amd.variants import Variant
    comm = job.init()
    ok = []
    for case in SEQS["matrices"]:
        graphs = LCSgraph.from_variant_batch(case["reference"], [[Variant(*v)] for v in case["variants"]])
        for tile in (8, 256):
            stats = {}
            labels = compare_matrix(case["reference"], graphs, tile=tile, comm=comm, stats=stats)
            ok.append(labels.tolist() == case["labels"] and "seconds_all_gather_assemble" in stats)
    comm.close()
    out.put((ok, "torch" in sys.modules))


def test_compare_matrix_all_gather_over_rccl():
    """The RCCL leg of the N > 1 path on the one GPU a test box has: a job of one rank, the library's own
    communicator (librccl bound at run time, algebra_amd/comm.py for the id), ncclAllGather from the buffer the tile
    kernels wrote into the buffer the assembly kernel reads (malcs_relmatrix_assemble_comm).  No torch in the process."""
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = 33500 + np.random.RandomState().randint(0, 2000)
    proc = ctx.Process(target=_matrix_rccl_rank, args=(port, out))
    proc.start()
    ok, torch_loaded = out.get(timeout=150)
    proc.join(timeout=60)
    assert proc.exitcode == 0
    assert ok and all(ok)
    assert not torch_loaded


def test_compare_matrix_c4_full_size():
    """BASELINE config 4 at the size bench.py times: 5 000 clustered variants in a 100 kb locus, all 12 497 500 pairs
    against what the reference returned for them (tests/golden/c4_matrix.json, made by make_golden_c4.py): supremals and
    distances of the 5 000 graphs, SHA-256 and histogram of the upper triangle, 3 000 explicit labels below the
    diagonal, the diagonal."""
    import hashlib
    from conftest import load_golden
    from util import digest
    from algebra_amd.lcs import LCSgraph
    from algebra_amd.relations import compare_matrix
    from algebra_amd.variants import Variant
    gold = load_golden("c4_matrix.json")
    case = gen.matrix_case(**gold["params"])
    assert digest([case["reference"], [list(v) for v in case["variants"]]]) == gold["input_sha256"]
    graphs = LCSgraph.from_variant_batch(case["reference"], [[Variant(*v)] for v in case["variants"]])
    assert digest([[g.supremal.start, g.supremal.end, g.supremal.sequence] for g in graphs]) == gold["supremals_sha256"]
    assert digest([g.distance for g in graphs]) == gold["distances_sha256"]
    labels = compare_matrix(case["reference"], graphs)
    n = gold["n"]
    assert labels.shape == (n, n)
    upper = labels[np.triu_indices(n, 1)]                              # row-major, i < j
    assert np.bincount(upper, minlength=5).tolist() == gold["upper_histogram"]
    assert hashlib.sha256(np.ascontiguousarray(upper, dtype=np.uint8).tobytes()).hexdigest() == gold["upper_sha256"]
    assert (np.diagonal(labels) == 0).all()                            # a graph is equivalent to itself
    for j, i, code in gold["below_diagonal_sample"]:
        assert labels[j, i] == code, (j, i)
