"""All-vs-all relation matrix over LCS graphs, tiled and sharded over GPUs.

Every cell is `graph_based.compare(reference, graphs[i], graphs[j])` (reference
algebra/relations/graph_based.py:33-79; the shape of scripts/benchmark.py:38-52 `pairwise`).
The O(1) pre-filters (:51-54) are evaluated for a whole tile at once with numpy; the pairs that
survive go to the GPU as one batch of edit distances (:65); the few that the distance arithmetic
(:67-74) cannot decide fall back to the atom-set test (:76-79).

Multi-GPU: upper-triangular tiles are dealt round-robin to the ranks of a torch.distributed
group (one process per GPU).  The only collective is one all-gather of the uint8 label tiles
(RCCL over xGMI with the "nccl" backend; "gloo" in the CPU tests).
"""

import numpy as np

from .graph_based import union_windows
from .relation import CODES, Relation

EQUIVALENT = CODES[Relation.EQUIVALENT]
CONTAINS = CODES[Relation.CONTAINS]
IS_CONTAINED = CODES[Relation.IS_CONTAINED]
OVERLAP = CODES[Relation.OVERLAP]
DISJOINT = CODES[Relation.DISJOINT]
_MIRROR = np.array([EQUIVALENT, IS_CONTAINED, CONTAINS, OVERLAP, DISJOINT], dtype=np.uint8)


def upper_tiles(n, tile):
    """Upper-triangular tile coordinates (block row, block col), row-major."""
    blocks = (n + tile - 1) // tile
    return [(bi, bj) for bi in range(blocks) for bj in range(bi, blocks)]


def shard(items, rank, world):
    """Round-robin deal of work items to ranks."""
    return items[rank::world]


class _Prepared:
    """Per-variant arrays for the vectorised pre-filters."""

    def __init__(self, graphs):
        sup = [g.supremal for g in graphs]
        self.start = np.array([v.start for v in sup], dtype=np.int64)
        self.end = np.array([v.end for v in sup], dtype=np.int64)
        self.distance = np.array([g.distance for g in graphs], dtype=np.int64)
        keys = {}
        self.key = np.array([keys.setdefault((v.start, v.end, v.sequence), len(keys)) for v in sup], dtype=np.int64)
        symbols = {}
        masks = []
        for v in sup:
            mask = 0
            for ch in set(v.sequence):
                mask |= 1 << symbols.setdefault(ch, len(symbols))
            masks.append(mask)
        self.wide = len(symbols) > 63
        self.masks = masks if self.wide else np.array(masks, dtype=np.uint64)
        self.atoms = [None] * len(graphs)

    def symbols_disjoint(self, rows, cols):
        if not self.wide:
            return (self.masks[rows][:, None] & self.masks[cols][None, :]) == 0
        return np.array([[self.masks[i] & self.masks[j] == 0 for j in cols] for i in rows], dtype=bool)


def _tile_prefilter(prep, rows, cols):
    """Labels decided without DP for the tile rows x cols; 255 marks undecided cells."""
    labels = np.full((len(rows), len(cols)), 255, dtype=np.uint8)
    s_r, e_r = prep.start[rows][:, None], prep.end[rows][:, None]
    s_c, e_c = prep.start[cols][None, :], prep.end[cols][None, :]
    overlapping = (s_c < e_r) & (s_r < e_c)
    apart = (s_c > e_r) | (s_r > e_c)
    disjoint = ~overlapping & (apart | prep.symbols_disjoint(rows, cols))       # Variant.is_disjoint, variant.py:126-133
    labels[disjoint] = DISJOINT
    labels[prep.key[rows][:, None] == prep.key[cols][None, :]] = EQUIVALENT     # graph_based.py:51 comes first
    return labels


def compare_matrix(reference, graphs, tile=256, group=None, distance_fn=None, stats=None):
    """uint8 matrix L with L[i, j] = code of compare(reference, graphs[i], graphs[j]).

    Codes follow the order of `Relation` (0 equivalent, 1 contains, 2 is_contained, 3 overlap,
    4 disjoint).  With an initialised torch.distributed process group (or `group`), the tiles are
    sharded over its ranks and all-gathered; every rank returns the full matrix.
    """
    if distance_fn is None:
        from ..lcs import edit_distance_batch as distance_fn
    n = len(graphs)
    rank, world, dist = 0, 1, None
    try:
        import torch.distributed as dist_mod
        if dist_mod.is_available() and dist_mod.is_initialized():
            dist = dist_mod
            rank, world = dist.get_rank(group), dist.get_world_size(group)
    except ImportError:
        pass

    prep = _Prepared(graphs)
    tiles = upper_tiles(n, tile)
    mine = shard(tiles, rank, world)
    out = np.zeros((len(mine), tile, tile), dtype=np.uint8)

    # pass 1: pre-filters per tile, collect the pairs that need the DP
    pending = []                                   # (tile index, r, c, i, j)
    for t, (bi, bj) in enumerate(mine):
        rows = np.arange(bi * tile, min(n, (bi + 1) * tile))
        cols = np.arange(bj * tile, min(n, (bj + 1) * tile))
        labels = _tile_prefilter(prep, rows, cols)
        if bi == bj:
            labels[np.tril_indices(len(rows), -1)] = EQUIVALENT      # lower half of a diagonal tile is mirrored later
        out[t, :len(rows), :len(cols)] = labels
        rr, cc = np.nonzero(labels == 255)
        pending.extend((t, int(r), int(c), int(rows[r]), int(cols[c])) for r, c in zip(rr, cc))

    # pass 2: one batch of edit distances over the union windows
    if pending:
        windows = [union_windows(reference, graphs[i].supremal, graphs[j].supremal) for _, _, _, i, j in pending]
        distance = np.asarray(distance_fn(windows), dtype=np.int64)
        li = prep.distance[[p[3] for p in pending]]
        rj = prep.distance[[p[4] for p in pending]]
        code = np.full(len(pending), 255, dtype=np.uint8)
        code[rj - li == distance] = IS_CONTAINED                    # graph_based.py:67-74, first match wins
        code[li - rj == distance] = CONTAINS
        code[li + rj == distance] = DISJOINT
        for k, (t, r, c, i, j) in enumerate(pending):
            if code[k] == 255:                                      # graph_based.py:76-79
                for idx in (i, j):
                    if prep.atoms[idx] is None:
                        prep.atoms[idx] = graphs[idx].uniq_atomics()
                code[k] = DISJOINT if prep.atoms[i].isdisjoint(prep.atoms[j]) else OVERLAP
            out[t, r, c] = code[k]
    if stats is not None:
        stats["dp_pairs"] = stats.get("dp_pairs", 0) + len(pending)
        stats["tiles"] = len(mine)

    # all-gather of the label tiles
    if world > 1:
        import torch
        per_rank = (len(tiles) + world - 1) // world
        backend = dist.get_backend(group)
        device = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
        send = torch.zeros((per_rank, tile, tile), dtype=torch.uint8, device=device)
        if len(mine):
            send[:len(mine)] = torch.from_numpy(out).to(device)
        recv = torch.empty((world * per_rank, tile, tile), dtype=torch.uint8, device=device)
        dist.all_gather_into_tensor(recv, send, group=group)
        gathered = recv.cpu().numpy().reshape(world, per_rank, tile, tile)
    else:
        gathered = out[None]

    full = np.zeros((n, n), dtype=np.uint8)
    for r in range(world):
        for t, (bi, bj) in enumerate(shard(tiles, r, world)):
            r0, c0 = bi * tile, bj * tile
            h, w = min(tile, n - r0), min(tile, n - c0)
            block = gathered[r, t, :h, :w]
            if bi == bj:
                upper = np.triu(block, 1)
                full[r0:r0 + h, c0:c0 + w] = upper + _MIRROR[upper].T * np.tri(h, w, -1, dtype=np.uint8)
            else:
                full[r0:r0 + h, c0:c0 + w] = block
                full[c0:c0 + w, r0:r0 + h] = _MIRROR[block].T
    np.fill_diagonal(full, EQUIVALENT)
    return full
