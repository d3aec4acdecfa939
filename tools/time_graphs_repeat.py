"""Ten calls of Batch.graphs() in a row on a C2-shaped batch: is any call of the series slower than the others? (dev tool)"""
import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import gen
from algebra_amd import _native as N
ctx = N.Context(0)
b = N.Batch(ctx, *gen.c2_batch(100000, seed=12), 4)
for _ in range(2):
    b.graphs().close()
walls = []
for _ in range(10):
    t0 = time.perf_counter(); g = b.graphs(); walls.append((time.perf_counter() - t0) * 1e3); g.close()
print("wall ms:", [round(w, 2) for w in walls])
