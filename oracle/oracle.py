"""ctypes wrapper of oracle/liboracle.so -- TEST INFRASTRUCTURE ONLY.

CPU restatement of the reference's LCS path (see lcs_oracle.c).  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg import this module; nothing under
algebra_amd/ does.
"""

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "liboracle.so")


def build():
    subprocess.check_call(["make", "-C", _HERE, "-s"])


def _load():
    if not os.path.exists(_LIB):
        build()
    lib = ctypes.CDLL(_LIB)
    c = ctypes
    vp, i64 = c.c_void_p, c.c_int64
    lib.orc_edit_distance.restype = i64
    lib.orc_edit_distance.argtypes = [vp, i64, vp, i64]
    lib.orc_trim.restype = None
    lib.orc_trim.argtypes = [vp, i64, vp, i64, c.POINTER(i64), c.POINTER(i64)]
    lib.orc_graph_create.restype = vp
    lib.orc_graph_create.argtypes = [vp, i64, vp, i64, i64, i64]
    lib.orc_graph_from_runs.restype = vp
    lib.orc_graph_from_runs.argtypes = [i64, i64, i64, vp, i64, i64, i64]
    lib.orc_graph_free.restype = None
    lib.orc_graph_free.argtypes = [vp]
    for name in ("status",):
        getattr(lib, f"orc_graph_{name}").restype = c.c_int
        getattr(lib, f"orc_graph_{name}").argtypes = [vp]
    for name in ("distance", "n_levels", "n_runs", "n_nodes", "n_edges"):
        getattr(lib, f"orc_graph_{name}").restype = i64
        getattr(lib, f"orc_graph_{name}").argtypes = [vp]
    for name in ("supremal", "runs", "nodes", "edges"):
        getattr(lib, f"orc_graph_{name}").restype = None
        getattr(lib, f"orc_graph_{name}").argtypes = [vp, vp]
    lib.orc_edit_distance_batch.restype = None
    lib.orc_edit_distance_batch.argtypes = [vp, vp, vp, vp, vp, c.c_uint64, vp]
    return lib


lib = _load()


def encode(*strings):
    """Map the distinct symbols of the given strings to uint8 codes (ACGT -> 0..3 first)."""
    table = {}
    for ch in "ACGT":
        table[ch] = len(table)
    out = []
    for s in strings:
        arr = np.empty(len(s), dtype=np.uint8)
        for i, ch in enumerate(s):
            if ch not in table:
                if len(table) >= 256:
                    raise ValueError("more than 256 distinct symbols")
                table[ch] = len(table)
            arr[i] = table[ch]
        out.append(arr)
    return out


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def edit_distance(a, b):
    """edit_distance(a, b) on str or uint8 arrays."""
    if isinstance(a, str):
        a, b = encode(a, b)
    return int(lib.orc_edit_distance(_p(a), len(a), _p(b), len(b)))


def trim(a, b):
    if isinstance(a, str):
        a, b = encode(a, b)
    p, s = ctypes.c_int64(), ctypes.c_int64()
    lib.orc_trim(_p(a), len(a), _p(b), len(b), ctypes.byref(p), ctypes.byref(s))
    return p.value, s.value


class Graph:
    """Result of LCSgraph(reference, observed, shift, max_distance) as plain arrays."""

    def __init__(self, handle):
        self.status = lib.orc_graph_status(handle)
        self.distance = int(lib.orc_graph_distance(handle))
        self.n_levels = int(lib.orc_graph_n_levels(handle))
        n_runs = int(lib.orc_graph_n_runs(handle))
        self.runs = np.zeros((n_runs, 4), dtype=np.int64)
        if n_runs:
            lib.orc_graph_runs(handle, _p(self.runs))
        self.supremal = None
        self.nodes = np.zeros((0, 3), dtype=np.int64)
        self.edges = np.zeros((0, 6), dtype=np.int64)
        if self.status == 0:
            sup = np.zeros(4, dtype=np.int64)
            lib.orc_graph_supremal(handle, _p(sup))
            self.supremal = tuple(int(x) for x in sup)
            nn, ne = int(lib.orc_graph_n_nodes(handle)), int(lib.orc_graph_n_edges(handle))
            self.nodes = np.zeros((nn, 3), dtype=np.int64)
            self.edges = np.zeros((ne, 6), dtype=np.int64)
            if nn:
                lib.orc_graph_nodes(handle, _p(self.nodes))
            if ne:
                lib.orc_graph_edges(handle, _p(self.edges))
        lib.orc_graph_free(handle)

    def level_lists(self):
        """The reference's `lcs_nodes` list of lists of (row, col, length)."""
        levels = [[] for _ in range(self.n_levels)]
        for row, col, length, level in self.runs.tolist():
            if 0 <= level < self.n_levels:
                levels[level].append((row, col, length))
        return levels


def lcs_graph(reference, observed, shift=0, max_distance=0):
    if isinstance(reference, str):
        reference, observed = encode(reference, observed)
    h = lib.orc_graph_create(_p(reference), len(reference), _p(observed), len(observed), int(shift), int(max_distance or 0))
    return Graph(h)


def graph_from_runs(m, n, shift, runs, n_levels, distance):
    runs = np.ascontiguousarray(runs, dtype=np.int64).reshape(-1, 4)
    h = lib.orc_graph_from_runs(int(m), int(n), int(shift), _p(runs), len(runs), int(n_levels), int(distance))
    return Graph(h)


def edit_distance_batch(syms, a_off, a_len, b_off, b_len):
    syms = np.ascontiguousarray(syms, dtype=np.uint8)
    a_off = np.ascontiguousarray(a_off, dtype=np.uint64)
    b_off = np.ascontiguousarray(b_off, dtype=np.uint64)
    a_len = np.ascontiguousarray(a_len, dtype=np.uint32)
    b_len = np.ascontiguousarray(b_len, dtype=np.uint32)
    out = np.empty(len(a_off), dtype=np.int32)
    lib.orc_edit_distance_batch(_p(syms), _p(a_off), _p(a_len), _p(b_off), _p(b_len), len(a_off), _p(out))
    return out
