"""Window class: kernel time against sequence length at a fixed number of pairs (intercept = launch, prologue, join)."""
import os, sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import gen
from algebra_amd import _native as N

ctx = N.Context(0)
pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 98304
for length in [int(x) for x in sys.argv[2:]] or [250, 500, 1000, 2000, 4000]:
    data = gen.c2_batch(pairs, seed=2, length=length, p=10.0 / length)
    b = N.Batch(ctx, *data, 4)
    info = b.plan_info()
    d = b.distances()
    b.time_dominant_kernel(5)
    k = b.time_dominant_kernel(50) / 50
    print(f"pairs={pairs} length={length} tiles={info['window_tiles']} two_limb={info['two_limb_tiles']} kernel={k*1e3:.1f} us "
          f"cells={info['window_cells']:.3e} mean_d={d.mean():.2f}", flush=True)
    b.close()
