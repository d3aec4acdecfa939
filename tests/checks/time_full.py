"""Time the full-matrix kernel of family 1 on the C2 batch (window class off) (dev tool)."""
import os, sys
os.environ["MALCS_DIAG_NLW"] = "0"
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import gen
from algebra_amd import _native as N
from oracle import oracle as O

ctx = N.Context(0)
for pairs in [int(x) for x in sys.argv[1:]] or [100000]:
    data = gen.c2_batch(pairs, seed=2)
    b = N.Batch(ctx, *data, 4)
    d = b.distances()
    b.time_distances(5)
    t = b.time_distances(30) / 30
    cells = b.plan_info()["full_cells"]
    pick = np.random.RandomState(0).choice(pairs, 2000, replace=False)
    ref = O.edit_distance_batch(data[0], data[1][pick], data[2][pick], data[3][pick], data[4][pick])
    print(f"pairs={pairs} full-matrix kernel {t*1e3:.1f} us  {cells/(t*1e-3)/1e12:.1f} TCUPS  "
          f"{cells*0.125/(t*1e-3)/1e12:.1f} T lane-op/s  ok={np.array_equal(d[pick], ref)}", flush=True)
    b.close()
