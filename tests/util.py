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


class HostTiles:
    """numpy restatement of kernel family 3 (csrc/malcs_relation.hip) with the interface of
    `_native.RelationMatrix` -- TEST INFRASTRUCTURE: the CPU tests inject it into `compare_matrix`
    (no GPU there), the GPU tests check the kernels against it."""

    MIRROR = None

    def __init__(self, start, end, key, symmask, tile, rank=0, world=1):
        import numpy as np
        self.np = np
        self.n, self.tile, self.rank, self.world = len(start), tile, rank, world
        blocks = (self.n + tile - 1) // tile
        self.all_tiles = [(bi, bj) for bi in range(blocks) for bj in range(bi, blocks)]
        self.mine = self.all_tiles[rank::world]
        self.per_rank = (len(self.all_tiles) + world - 1) // world
        self.labels = np.zeros((self.per_rank, tile, tile), dtype=np.uint8)
        self.cells = []
        n = self.n
        for t, (bi, bj) in enumerate(self.mine):
            rows = np.arange(bi * tile, min(n, (bi + 1) * tile))
            cols = np.arange(bj * tile, min(n, (bj + 1) * tile))
            s_r, e_r = start[rows][:, None], end[rows][:, None]
            s_c, e_c = start[cols][None, :], end[cols][None, :]
            overlapping = (s_c < e_r) & (s_r < e_c)
            apart = (s_c > e_r) | (s_r > e_c)
            shared = (symmask[rows][:, None] & symmask[cols][None, :]) != 0
            lab = np.full((len(rows), len(cols)), 255, dtype=np.uint8)
            lab[~overlapping & (apart | ~shared)] = 4
            lab[key[rows][:, None] == key[cols][None, :]] = 0
            if bi == bj:
                lab[np.tril_indices(len(rows), -1)] = 0
            self.labels[t, :len(rows), :len(cols)] = lab
            for r, c in zip(*np.nonzero(lab == 255)):
                self.cells.append((t, int(r), int(c), int(rows[r]), int(cols[c])))

    def pending(self):
        return self.np.array([[i, j] for *_, i, j in self.cells], dtype=self.np.int64).reshape(-1, 2)

    def resolve(self, codes):
        for (t, r, c, _, _), code in zip(self.cells, codes):
            self.labels[t, r, c] = code

    def tiles(self):
        return self.labels.copy()

    def assemble(self, gathered=None):
        np = self.np
        mirror = np.array([0, 2, 1, 3, 4], dtype=np.uint8)
        n, tile, world = self.n, self.tile, self.world
        blocks = self.labels[None] if gathered is None else np.asarray(gathered).reshape(world, self.per_rank, tile, tile)
        full = np.zeros((n, n), dtype=np.uint8)
        for r in range(world):
            for t, (bi, bj) in enumerate(self.all_tiles[r::world]):
                r0, c0 = bi * tile, bj * tile
                h, w = min(tile, n - r0), min(tile, n - c0)
                block = blocks[r, t, :h, :w]
                if bi == bj:
                    upper = np.triu(block, 1)
                    full[r0:r0 + h, c0:c0 + w] = upper + mirror[upper].T * np.tri(h, w, -1, dtype=np.uint8)
                else:
                    full[r0:r0 + h, c0:c0 + w] = block
                    full[c0:c0 + w, r0:r0 + h] = mirror[block].T
        np.fill_diagonal(full, 0)
        return full

    def close(self):
        pass


def gloo_gather(dist):
    """`gather` for compare_matrix over an initialised torch.distributed group (TEST INFRASTRUCTURE: the product's own
    collective is RCCL inside the library; gloo lets two ranks meet on a CPU or on one GPU)."""
    import numpy as np
    import torch

    def gather(block):
        send = torch.from_numpy(np.ascontiguousarray(block))
        recv = torch.empty((dist.get_world_size() * block.shape[0],) + tuple(block.shape[1:]), dtype=torch.uint8)
        dist.all_gather_into_tensor(recv, send)
        return recv.numpy()
    return gather
