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
    g.close()
    t3 = time.perf_counter()
    print("library call %.2f ms, columns read out %.2f ms, close %.2f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3), flush=True)
