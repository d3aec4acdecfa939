"""All-vs-all relation matrix over LCS graphs, tiled and sharded over GPUs.

Every cell is `graph_based.compare(reference, graphs[i], graphs[j])` (reference
algebra/relations/graph_based.py:33-79; the shape of scripts/benchmark.py:38-52 `pairwise`).
The O(1) pre-filters (:51-54) are evaluated for all tiles of a rank by kernel family 3
(`malcs_relmatrix_*`, csrc/malcs_relation.hip); the pairs that survive go to the GPU as one batch of
edit distances (:65, kernel family 1); the few that the distance arithmetic (:67-74) cannot decide
look at atom masks (:76-79); the device scatters the verdicts into the label tiles and assembles
the full matrix.

Multi-GPU: upper-triangular tiles are dealt round-robin to the ranks of a job (one process per GPU).  The only
collective is one all-gather of the uint8 label tiles: RCCL over xGMI, called by the library itself on its own
buffers (`comm` = `algebra_amd.comm.init()`, malcs_relmatrix_assemble_comm).  Nothing here imports torch; a caller
with another transport passes `rank`, `world` and a `gather` function (the CPU tests do, with gloo).
"""

import numpy as np

from .graph_based import union_windows
from .relation import CODES, Relation

EQUIVALENT = CODES[Relation.EQUIVALENT]
CONTAINS = CODES[Relation.CONTAINS]
IS_CONTAINED = CODES[Relation.IS_CONTAINED]
OVERLAP = CODES[Relation.OVERLAP]
DISJOINT = CODES[Relation.DISJOINT]


def upper_tiles(n, tile):
    """Upper-triangular tile coordinates (block row, block col), row-major."""
    blocks = (n + tile - 1) // tile
    return [(bi, bj) for bi in range(blocks) for bj in range(bi, blocks)]


def shard(items, rank, world):
    """Round-robin deal of work items to ranks."""
    return items[rank::world]


class Prepared:
    """Per-variant arrays for the pre-filters: interval, distance, a number per distinct supremal
    (equal supremals <=> equal keys, graph_based.py:51) and one bit per symbol of the inserted
    sequence (Variant.is_disjoint, variants/variant.py:133).  Beyond 64 distinct symbols the bits
    are folded (`folded`): a shared bit then no longer proves a shared symbol, so the undecided
    pairs are re-checked exactly before they go to the DP."""

    def __init__(self, graphs):
        sup = [g.supremal for g in graphs]
        self.supremal = sup
        self.start = np.array([v.start for v in sup], dtype=np.int64)
        self.end = np.array([v.end for v in sup], dtype=np.int64)
        self.distance = np.array([g.distance for g in graphs], dtype=np.int64)
        keys = {}
        self.key = np.array([keys.setdefault((v.start, v.end, v.sequence), len(keys)) for v in sup], dtype=np.uint32)
        symbols = {}
        masks = []
        for v in sup:
            mask = 0
            for ch in set(v.sequence):
                mask |= 1 << (symbols.setdefault(ch, len(symbols)) % 64)
            masks.append(mask)
        self.folded = len(symbols) > 64
        self.symmask = np.array(masks, dtype=np.uint64)


_DNA = np.full(256, 255, dtype=np.uint8)
_DNA[[ord(ch) for ch in "ACGT"]] = np.arange(4, dtype=np.uint8)


def _dna_codes(text):
    """str -> codes 0 ... 3, or None when it holds anything but A, C, G, T."""
    try:
        codes = _DNA[np.frombuffer(text.encode("latin-1"), dtype=np.uint8)]
    except UnicodeEncodeError:
        return None
    return codes if (len(codes) == 0 or codes.max() < 4) else None


def device_inputs(reference, graphs, prep):
    """What `malcs_relmatrix_decide` needs to settle the undecided pairs on the device, or None when it does not
    apply (graphs put together by hand, symbols other than A, C, G, T, folded symbol masks): reference and supremal
    sequences as codes, the graphs' distances, their atom masks (computed by the library from the edge arrays)."""
    if prep.folded:
        return None
    ref = _dna_codes(reference)
    if ref is None:
        return None
    sup = _dna_codes("".join(v.sequence for v in prep.supremal))
    if sup is None:
        return None
    sup_off = np.zeros(len(graphs) + 1, dtype=np.uint64)
    np.cumsum([len(v.sequence) for v in prep.supremal], out=sup_off[1:])
    atom_off = np.zeros(len(graphs) + 1, dtype=np.uint64)
    np.cumsum(prep.end - prep.start + 1, out=atom_off[1:])
    atoms = np.zeros(int(atom_off[-1]), dtype=np.uint8)
    by_handle = {}
    for k, g in enumerate(graphs):
        result = g.__dict__.get("_result")
        if result is None:
            return None
        by_handle.setdefault(id(result[0]), (result[0], [], []))
        by_handle[id(result[0])][1].append(k)
        by_handle[id(result[0])][2].append(result[1])
    for handle, where, index in by_handle.values():
        observed = [graphs[k]._observed for k in where]
        obs = _dna_codes("".join(observed))
        if obs is None:
            return None
        obs_off = np.zeros(len(where) + 1, dtype=np.uint64)
        np.cumsum([len(x) for x in observed], out=obs_off[1:])
        where = np.asarray(where)
        handle.atoms(index, obs, obs_off, np.stack([atom_off[where], atom_off[where + 1]], 1), atoms)
    return ref, sup, sup_off, prep.distance, atoms, atom_off


def columns_of(reference, graphs):
    """The per-variant columns of a matrix: (`Prepared`, `device_inputs` or None).  A `GraphBatch` keeps them
    (`GraphBatch.columns`), so the second matrix over the same batch finds them made."""
    prep = Prepared(graphs)
    return prep, device_inputs(reference, graphs, prep)


class LibraryPrepared:
    """`Prepared` over a `VariantGraphBatch`: the same columns, as malcs_vgraphs_matrix_columns made them; the list of
    supremal `Variant`s only when somebody asks (the paths that decide pairs in Python)."""

    def __init__(self, batch, key, symmask, folded):
        native = batch._native
        self._batch = batch
        self.start = np.ascontiguousarray(native.supremal[:, 0])
        self.end = np.ascontiguousarray(native.supremal[:, 1])
        self.distance = native.distance
        self.key, self.symmask, self.folded = key, symmask, folded

    @property
    def supremal(self):
        made = self.__dict__.get("_supremal")
        if made is None:
            made = self.__dict__["_supremal"] = [g.supremal for g in self._batch]
        return made


def columns_from_library(batch):
    """`columns_of` for a `VariantGraphBatch` over the reference it was made on."""
    native = batch._native
    key, symmask, folded, sup_off, sup, atom_off, atoms = native.matrix_columns(with_atoms=batch._dna)
    prep = LibraryPrepared(batch, key, symmask, folded)
    on_device = (batch._ref_codes, sup, sup_off, prep.distance, atoms, atom_off) if (batch._dna and not folded) else None
    return prep, on_device


def compare_matrix(reference, graphs, tile=256, comm=None, rank=0, world=1, gather=None, distance_fn=None, stats=None,
                   tiles_cls=None):
    """uint8 matrix L with L[i, j] = code of compare(reference, graphs[i], graphs[j]).

    Codes follow the order of `Relation` (0 equivalent, 1 contains, 2 is_contained, 3 overlap,
    4 disjoint).  In a job of several ranks the tiles are sharded over them and all-gathered; every rank
    returns the full matrix.  `comm` is this rank's `_native.Comm` (`algebra_amd.comm.init()`): the library runs the
    all-gather over RCCL on its own device buffers (also in a job of one rank, where it is a copy).  Without a
    communicator, `rank` / `world` say whose tiles to compute and `gather(block)` exchanges them: it takes this
    rank's (tiles_per_rank, tile, tile) uint8 block and returns every rank's, stacked in rank order.

    `distance_fn` and `tiles_cls` exist for the host-logic tests (CPU): they inject
    distances computed elsewhere and a numpy restatement of the tile kernels.  The product path
    leaves both unset and runs on the device.
    """
    if distance_fn is None:
        from ..lcs import edit_distance_batch as distance_fn
    n = len(graphs)
    if comm is not None:
        rank, world = comm.rank, comm.world
    elif world > 1 and gather is None:
        raise ValueError("a job of several ranks needs a communicator or a gather function")

    import time
    clock = [time.perf_counter()]

    def lap(name):
        if stats is not None:
            now = time.perf_counter()
            stats["seconds_" + name] = stats.get("seconds_" + name, 0.0) + now - clock[0]
            clock[0] = now

    kept = None
    if tiles_cls is None and hasattr(graphs, "columns"):        # a GraphBatch: the columns are made once per batch
        kept = graphs.columns(reference, columns_of)
    prep = kept[0] if kept else Prepared(graphs)
    lap("prepare")
    if tiles_cls is None:
        from .. import _native
        tiles = _native.RelationMatrix(_native.default_context(), prep.start, prep.end, prep.key, prep.symmask, tile, rank, world)
    else:
        tiles = tiles_cls(prep.start, prep.end, prep.key, prep.symmask, tile, rank, world)
    try:
        # the pairs the pre-filters leave: on the device when the library can do all of it there (union windows, one
        # batch of distances, distance arithmetic, atom masks -- malcs_relmatrix_decide), else one batch of edit
        # distances over union windows put together here and the verdicts in numpy / Python
        if kept:
            on_device = kept[1]
        else:
            on_device = device_inputs(reference, graphs, prep) if (tiles_cls is None and hasattr(tiles, "decide")) else None
        if on_device is not None:
            lap("device_inputs")
            n_dp = tiles.decide(*on_device, 4)
            lap("decide")
            pending = ()
            if stats is not None:
                stats["dp_pairs"] = stats.get("dp_pairs", 0) + n_dp
        else:
            pending = tiles.pending()
            lap("tiles")
        if len(pending):
            li_idx, rj_idx = pending[:, 0], pending[:, 1]
            code = np.full(len(pending), 255, dtype=np.uint8)
            todo = np.arange(len(pending))
            if prep.folded:
                exact = np.array([prep.supremal[i].is_disjoint(prep.supremal[j]) for i, j in pending.tolist()], dtype=bool)
                code[exact] = DISJOINT
                todo = todo[~exact]
            if len(todo):
                windows = [union_windows(reference, prep.supremal[i], prep.supremal[j]) for i, j in pending[todo].tolist()]
                lap("windows")
                distance = np.asarray(distance_fn(windows), dtype=np.int64)
                lap("distances")
                li, rj = prep.distance[li_idx[todo]], prep.distance[rj_idx[todo]]
                sub = np.full(len(todo), 255, dtype=np.uint8)
                sub[rj - li == distance] = IS_CONTAINED                 # graph_based.py:67-74, first match wins
                sub[li - rj == distance] = CONTAINS
                sub[li + rj == distance] = DISJOINT
                for k in np.nonzero(sub == 255)[0].tolist():             # graph_based.py:76-79
                    i, j = pending[todo[k]].tolist()
                    sub[k] = DISJOINT if graphs[i].is_disjoint(graphs[j]) else OVERLAP
                code[todo] = sub
            tiles.resolve(code)
            lap("verdicts")
        if stats is not None:
            stats["dp_pairs"] = stats.get("dp_pairs", 0) + len(pending)
            stats["tiles"] = len(shard(upper_tiles(n, tile), rank, world))
            stats["decided_on_device"] = on_device is not None

        # all-gather of the label tiles
        if comm is not None:
            full = tiles.assemble_comm(comm)
            lap("all_gather_assemble")
            return full
        gathered = None
        if world > 1:
            gathered = np.ascontiguousarray(gather(tiles.tiles()), dtype=np.uint8)
            lap("all_gather")
        full = tiles.assemble(gathered)
        lap("assemble")
        return full
    finally:
        tiles.close()
