"""Sizes of the graphs of a C2-shaped batch: runs, nodes, edges per pair (dev tool; what the LDS arenas must hold)."""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import gen
from algebra_amd import _native as N
ctx = N.Context(0)
b = N.Batch(ctx, *gen.c2_batch(100000, seed=12), 4)
g = b.graphs()
c = g.counts
print("columns:", c[:3].tolist())
for k in range(c.shape[1]):
    col = c[:, k]
    print(k, "mean %.1f" % col.mean(), "percentiles 50/90/99/99.9/max:", [int(np.percentile(col, p)) for p in (50, 90, 99, 99.9, 100)])
