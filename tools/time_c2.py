"""Time family 1 on C2 with the library given in MALCS_LIB (dev tool)."""
import os, sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import gen
from algebra_amd import _native as N
ctx = N.Context(0)
n = int(os.environ.get("PAIRS", "100000"))
data = gen.c2_batch(n)
bt = N.Batch(ctx, *data, 4)
d = bt.distances()
cells = float((data[2].astype(np.float64) * data[4]).sum())
best = min(bt.time_distances(20) / 20 for _ in range(5))
print(f"{os.environ.get('TAG','')}: {best:.4f} ms/launch {cells/best/1e9:.1f} TCUPS checksum {int(d.sum())}")
