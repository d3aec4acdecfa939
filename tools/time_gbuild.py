import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gen
from algebra_amd import _native as N
ctx = N.Context(0)
b = N.Batch(ctx, *gen.c2_batch(100000, seed=12), 4)
for _ in range(2): b.graphs().close()
import time
t0 = time.perf_counter(); g = b.graphs(); wall = (time.perf_counter() - t0) * 1e3; print("wall_ms", round(wall, 2), os.environ.get("MALCS_LIB", "default"), "device_build_ms", round(float(g.device_build_ms), 3), "built", int(g.device_built), "backward_sweep_ms", round(float(g.timing[0]), 3), "forward_sweep_ms", round(float(g.timing[1]), 3)); t0 = time.perf_counter(); g.close(); print("close_ms", round((time.perf_counter() - t0) * 1e3, 2))
b.close()
