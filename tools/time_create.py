"""Host-to-result time of one C2 batch through the C-ABI (dev tool): batch creation (intern, H2D, pack, plan,
layout) and the distance call (launch + D2H), separately.  python tools/time_create.py [pairs]"""
import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import gen
from algebra_amd import _native as N

pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
ctx = N.Context(0)
data = gen.c2_batch(pairs, seed=2)
N.Batch(ctx, *gen.c2_batch(1000, seed=3), 4).close()          # code objects, first allocations
for rep in range(4):
    t0 = time.perf_counter()
    b = N.Batch(ctx, *data, 4)
    t1 = time.perf_counter()
    d = b.distances()
    t2 = time.perf_counter()
    b.close()
    print(f"pairs={pairs} create={1e3 * (t1 - t0):.1f} ms distances={1e3 * (t2 - t1):.2f} ms "
          f"-> {pairs / (t2 - t0):.3e} pairs/s from host bytes ({data[0].nbytes / 1e6:.0f} MB), sum={int(d.sum())}", flush=True)
b = N.Batch(ctx, *data, 4)
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(10):
        b.replan()
        b.launch_distances()
    ctx.sync()
    t1 = time.perf_counter()
    print(f"plan + launch on resident planes: {1e2 * (t1 - t0):.3f} ms per batch -> {pairs * 10 / (t1 - t0):.3e} pairs/s", flush=True)
