import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, gen
from algebra_amd import _native as N
b = N.Batch(N.default_context(), *gen.c2_batch(100000, seed=2), 4)
for _ in range(3): b.replan()
info = b.plan_info(); count, nt = info["window_pairs"], info["window_tiles"]
pid = b.plan_dump(1).reshape(nt * 64, 16)[:count, 9].astype(np.int64)
t = pid[: (count // 64) * 64].reshape(-1, 64)
spread = t.max(1) - t.min(1)
print("pair-id spread within a tile: median %d mean %d p90 %d max %d" % (np.median(spread), spread.mean(), np.percentile(spread, 90), spread.max()))
print("inversions between consecutive slots (fraction):", float((np.diff(pid) < 0).mean()))
b.close()
