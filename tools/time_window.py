"""Time the window class of family 1 on C2-shaped batches (dev tool): python tools/time_window.py [pairs ...]"""
import os, sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import gen
from algebra_amd import _native as N

ctx = N.Context(0)
for pairs in [int(x) for x in sys.argv[1:]] or [100000]:
    data = gen.c2_batch(pairs, seed=2)
    b = N.Batch(ctx, *data, 4)
    info = b.plan_info()
    d = b.distances()
    b.time_dominant_kernel(5)
    k = b.time_dominant_kernel(50) / 50
    s = b.time_distances(50) / 50
    print(f"pairs={pairs} nlw={info['window_limbs']} kernel={k*1e3:.1f} us step={s*1e3:.1f} us "
          f"pairs/s={pairs/(s*1e-3):.3e} exec={info['window_cells']*0.125/(k*1e-3)/1e12:.1f} T lane-op/s "
          f"mean_d={d.mean():.3f} sum={int(d.sum())}", flush=True)
    b.close()
