"""Time family 2 (graphs) on C2-shaped pairs (dev tool)."""
import os, sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import gen
from algebra_amd import _native as N
ctx = N.Context(0)
n = int(os.environ.get("PAIRS", "20000"))
data = gen.c2_batch(n)
bt = N.Batch(ctx, *data, 4)
cells = float((data[2].astype(np.float64) * data[4]).sum())
for rep in range(3):
    t0 = time.time()
    g = bt.graphs()
    t1 = time.time()
    tm = g.timing
    print(f"pairs={n} wall={t1-t0:.3f}s fwd={tm[0]:.2f}ms reach={tm[1]:.2f}ms host={tm[2]:.1f}ms  "
          f"fwd {cells*0.25/tm[0]/1e6:.0f} GB/s reach {cells*0.25/tm[1]/1e6:.0f} GB/s  nodes/pair={g.counts[:,2].mean():.1f} runs/pair={g.counts[:,0].mean():.1f}", flush=True)
    g.close()
