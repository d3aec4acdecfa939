import os, sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, gen
from algebra_amd import _native as N
ctx = N.Context(0)
for pairs, length, p, seed in [(100000, 1000, 0.01, 2), (6000, 1000, 0.01, 21), (6000, 1000, 0.02, 22), (5000, 700, 0.04, 23), (5000, 600, 0.0, 24), (20000, 2000, 0.005, 5), (20000, 400, 0.03, 6)]:
    data = gen.c2_batch(pairs, seed=seed, length=length, p=p)
    b = N.Batch(ctx, *data, 4)
    info = b.plan_info()
    d = b.distances()
    keys = [k for k in info if "sample" in k or "two_limb" in k or k == "window_tiles"]
    print(os.environ.get("MALCS_SAMPLE_WAVE", "lanes"), pairs, length, p, "true mean %.1f sd %.1f max %d" % (d.mean(), d.std(), d.max()), {k: info[k] for k in keys})
    b.close()
