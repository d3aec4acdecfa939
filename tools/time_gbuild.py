import os, sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import gen
from algebra_amd import _native as N
ctx = N.Context(0)
b = N.Batch(ctx, *gen.c2_batch(100000, seed=12), 4)
for _ in range(2): b.graphs().close()
g = b.graphs(); print(os.environ.get("MALCS_LIB", "default"), "device_build_ms", round(float(g.device_build_ms), 3), "built", int(g.device_built)); g.close()
b.close()
