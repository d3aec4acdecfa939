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


def check_window_plan(batch):
    """What the plan of a batch's window class (made on the device, malcs_plan.inc) must satisfy, read back through
    malcs_batch_plan_dump: the slots hold every pair of the class once, sorted by (group, blocks); the pieces of a tile cover
    its blocks exactly and their rows do not overlap; every workgroup's table entry is its piece's place; the row buffer
    holds them all.  Returns a short summary."""
    import numpy as np
    info = batch.plan_info()
    if not info["window_pairs"]:
        return {"window_pairs": 0}
    count, n_tiles, L = info["window_pairs"], info["window_tiles"], info["workgroups"]
    item = batch.plan_dump(0).reshape(count, 16)
    sitem = batch.plan_dump(1).reshape(n_tiles * 64, 16)
    lay_off, lay_info = batch.plan_dump(2), batch.plan_dump(3)
    piece_off, piece_info, piece_join, piece_place = (batch.plan_dump(k) for k in (4, 5, 6, 7))
    PID, ITERS, DELTA = 9, 8, 7
    # slots: a permutation of the pairs, empty slots only at the end of the last tile
    used = sitem[:count]
    assert (sitem[count:, PID] == 0xffffffff).all(), "empty slots"
    assert (used[:, PID] != 0xffffffff).all()
    assert len(np.unique(used[:, PID])) == count, "a pair sits in two slots"
    key_item = {tuple(r[:9]) for r in item[:0]}                      # (records are compared through sorting below)
    a = np.sort(item[:, :9].view([("", item.dtype)] * 9), axis=0)
    b = np.sort(np.ascontiguousarray(used[:, :9]).view([("", used.dtype)] * 9), axis=0)
    assert (a == b).all(), "the sorted records are not the records"
    two = L == 2 * n_tiles
    half_mode = info["two_limb_tiles"] > 0
    grp = (used[:, DELTA] > 48).astype(np.int64) if half_mode else np.zeros(count, np.int64)
    order_key = grp * (1 << 20) + used[:, ITERS].astype(np.int64) if half_mode else used[:, ITERS].astype(np.int64) * 2 + (used[:, DELTA] > 48)
    assert (np.diff(order_key) >= 0).all(), "slots are not sorted"
    # tiles and their pieces
    blocks_mask = (1 << 28) - 1
    rows_fixed = 4 + (info["window_limbs"] - 1) * 2                    # (DNA: two planes; the check is used with DNA batches)
    spans = []
    n_pieces = 0
    for t in range(n_tiles):
        lo, hi = 64 * t, min(64 * t + 64, count)
        longest = int(used[lo:hi, ITERS].max())
        places = [2 * t, 2 * t + 1] if two else [t]
        got = [(int(lay_info[j]) & blocks_mask, int(lay_info[j]) >> 28, int(lay_off[j])) for j in places if lay_info[j]]
        assert got, t
        assert sum(g[0] for g in got) == longest, ("blocks of tile", t, got, longest)
        if len(got) == 2:
            assert got[0][1] & 0x6 == 0x4 and got[1][1] & 0x6 == 0x6, "flags of a cut tile"     # joined (| back)
            assert (got[0][1] ^ got[1][1]) & 0x9 == 0, "both parts of a tile run in the same mode"
            assert min(g[0] for g in got) >= 2
        for blocks, _, off in got:
            spans.append((off, off + rows_fixed + blocks * 4))
        n_pieces += len(got)
    spans.sort()
    for (a0, a1), (b0, b1) in zip(spans, spans[1:]):
        assert a1 <= b0, "rows of two pieces overlap"
    assert n_pieces == info["pieces"]
    # the table of the launch: every piece once
    live = piece_info != 0
    assert int(live.sum()) == n_pieces
    places = piece_place[live]
    assert len(np.unique(places)) == n_pieces
    assert (piece_info[live] == lay_info[places]).all() and (piece_off[live] == lay_off[places]).all()
    assert (piece_join[live] == (places >> 1 if two else places)).all()
    # loads of the SIMDs (workgroups w, w + S, ... share one)
    S = 1024
    cost = (piece_info & blocks_mask).astype(np.int64) * np.where(piece_info >> 31, 3, 4)
    load = np.zeros(S, np.int64)
    np.add.at(load, np.arange(L) % S, cost)
    return {"pieces": n_pieces, "tiles": n_tiles, "rows": spans[-1][1], "two_limb_tiles": info["two_limb_tiles"],
            "simd_load_max": int(load.max()), "simd_load_mean": float(load.mean()), "reported_worst": info["worst_simd"]}


def pack_arena(syms, offs, lens, n_planes, junk_tails=None):
    """The packed arena of include/malcs.h (malcs_batch_create_packed), made with numpy from the bytes: sequence s =
    syms[offs[s] : offs[s] + lens[s]] becomes ceil(len / 32) blocks of `n_planes` dwords, bit j of dword p of block k =
    bit p of symbol 32 k + j.  Returns (planes, dword offset of every sequence).  `junk_tails`: a RandomState that fills
    the bits behind every sequence's last symbol (they may hold anything)."""
    import numpy as np
    syms = np.asarray(syms, dtype=np.uint8)
    offs = np.asarray(offs, dtype=np.int64)
    lens = np.asarray(lens, dtype=np.int64)
    blocks = (lens + 31) // 32
    first = np.concatenate([[0], np.cumsum(blocks)])                   # first block of every sequence
    codes = np.zeros(int(first[-1]) * 32, dtype=np.uint8)
    if junk_tails is not None:
        codes[:] = junk_tails.randint(0, 1 << n_planes, size=codes.size)
    seq_of = np.repeat(np.arange(len(lens)), lens)                     # for every symbol: its sequence
    within = np.arange(int(lens.sum())) - np.repeat(np.cumsum(lens) - lens, lens)
    codes[first[seq_of] * 32 + within] = syms[offs[seq_of] + within]
    planes = np.empty((int(first[-1]), n_planes), dtype=np.uint32)
    for p in range(n_planes):
        bits = (codes >> p) & 1
        planes[:, p] = np.packbits(bits, bitorder="little").view("<u4")
    return planes.ravel(), (first[:-1] * n_planes).astype(np.uint32)


_hip = None


def hip_runtime():
    """The HIP runtime libmalcs is linked against, through ctypes (hipMalloc / hipMemcpy / hipFree): device memory for a
    test without a second framework initialising the GPU in this process."""
    global _hip
    if _hip is None:
        import ctypes
        from algebra_amd import _native  # noqa: F401  (loads libmalcs and with it the runtime; the soname then resolves to that copy)
        _hip = ctypes.CDLL("libamdhip64.so.7")
        _hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
        _hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
        _hip.hipFree.argtypes = [ctypes.c_void_p]
    return _hip


def to_device(array):
    """A copy of the numpy array in device memory: a ctypes.c_void_p the caller frees with hip_runtime().hipFree."""
    import ctypes
    hip = hip_runtime()
    p = ctypes.c_void_p()
    assert hip.hipMalloc(ctypes.byref(p), max(array.nbytes, 4)) == 0
    assert hip.hipMemcpy(p, array.ctypes.data_as(ctypes.c_void_p), array.nbytes, 1) == 0      # hipMemcpyHostToDevice
    return p
