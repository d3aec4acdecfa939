"""Dev tool: plan + launch per batch as the bench's headline times it (C2, 100 000 pairs), a few repeats:
    python tools/time_step.py [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, gen
from algebra_amd import _native as N
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
ctx = N.Context(0)
b = N.Batch(ctx, *gen.c2_batch(100000, seed=2), 4)
b.distances()
out = []
for rep in range(5):
    for _ in range(50):
        b.replan(); b.launch_distances()
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        b.replan(); b.launch_distances()
    ctx.sync()
    out.append((time.perf_counter() - t0) / steps * 1e6)
print("MALCS_DIAG_PAD", os.environ.get("MALCS_DIAG_PAD", "default"), "us per step:", " ".join("%.2f" % x for x in out), "min %.2f" % min(out))
b.close()
