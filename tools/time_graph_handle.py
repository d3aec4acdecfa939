"""Where the wall time of Batch.graphs() goes outside the library call: the C call itself against the read-out of the
per-pair columns (counts, status, distance, supremal) into fresh numpy arrays (dev tool)."""
import ctypes, sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import gen
from algebra_amd import _native as N
ctx = N.Context(0)
b = N.Batch(ctx, *gen.c2_batch(100000, seed=12), 4)
for _ in range(3):
    b.graphs().close()
for _ in range(4):
    h = ctypes.c_void_p()
    t0 = time.perf_counter()
    N.check(N.lib.malcs_lcs_graph_batch(b.handle, None, None, ctypes.byref(h)))
    t1 = time.perf_counter()
    g = N.Graphs(h, b.n_pairs)
    t2 = time.perf_counter()
    counts = np.zeros((b.n_pairs, 4), dtype=np.int64); status = np.zeros(b.n_pairs, dtype=np.int32)
    distance = np.zeros(b.n_pairs, dtype=np.int64); supremal = np.zeros((b.n_pairs, 4), dtype=np.int64)
    N.check(N.lib.malcs_graphs_counts(h, N.ptr(counts))); N.check(N.lib.malcs_graphs_status(h, N.ptr(status), N.ptr(distance)))
    N.check(N.lib.malcs_graphs_supremal(h, N.ptr(supremal)))
    assert np.array_equal(counts, g.counts) and np.array_equal(status, g.status) and np.array_equal(distance, g.distance) and np.array_equal(supremal, g.supremal)
    t2b = time.perf_counter()
    g.close()
    t3 = time.perf_counter()
    print("library call %.2f ms, columns read out %.2f ms, close %.2f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2b) * 1e3), flush=True)
# the pieces of the read-out
h = ctypes.c_void_p()
N.check(N.lib.malcs_lcs_graph_batch(b.handle, None, None, ctypes.byref(h)))
n = b.n_pairs
for _ in range(3):
    t0 = time.perf_counter()
    arrays = [N.pinned_empty((n, 4), dtype=np.int64), N.pinned_empty(n, dtype=np.int32), N.pinned_empty(n, dtype=np.int64), N.pinned_empty((n, 4), dtype=np.int64)]
    t1 = time.perf_counter()
    N.check(N.lib.malcs_graphs_columns(h, *[N.ptr(a) for a in arrays]))
    t2 = time.perf_counter()
    N.check(N.lib.malcs_graphs_columns(h, *[N.ptr(a) for a in arrays]))
    t3 = time.perf_counter()
    print("pinned arrays %.3f ms, columns %.3f ms, columns again %.3f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3), flush=True)
    del arrays
N.lib.malcs_graphs_destroy(h)
