"""Plan + launch per batch on resident planes, and what the plan decided (dev tool; MALCS_TRACE=1 for the host stages)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import gen
from algebra_amd import _native as N
ctx = N.Context(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
data = gen.c2_batch(n, seed=2)
b = N.Batch(ctx, *data, 4)
print(b.plan_info(), file=sys.stderr)
d0 = b.distances()
for _ in range(50):
    b.replan(); b.launch_distances()
ctx.sync()
for reps in (1, 20, 200):
    t0 = time.perf_counter()
    for _ in range(reps):
        b.replan(); b.launch_distances()
    t1 = time.perf_counter(); ctx.sync(); t2 = time.perf_counter()
    print("%d x (replan + launch): enqueue %.3f ms, total %.3f ms per batch" % (reps, (t1 - t0) * 1e3 / reps, (t2 - t0) * 1e3 / reps), file=sys.stderr, flush=True)
assert np.array_equal(b.fetch_distances(), d0)
each = b.kernel_durations(500)
print("window kernel: mean %.4f median %.4f min %.4f ms" % (each.mean(), np.median(each), each.min()), file=sys.stderr)
print(b.plan_info(), file=sys.stderr)
b.close()
