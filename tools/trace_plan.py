"""Stage times of the window plan on resident planes (MALCS_TRACE=1 python tools/trace_plan.py; dev tool)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gen
from algebra_amd import _native as N
ctx = N.Context(0)
b = N.Batch(ctx, *gen.c2_batch(100000, seed=2), 4)
for _ in range(5):
    b.replan(); b.launch_distances()
ctx.sync()
for _ in range(3):
    t0 = time.perf_counter(); b.replan(); t1 = time.perf_counter(); b.launch_distances(); ctx.sync(); t2 = time.perf_counter()
    print("replan %.3f ms, launch + sync %.3f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3), file=sys.stderr, flush=True)
b.close()
