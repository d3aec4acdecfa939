"""Two-limb mode of the window class (MALCS_DIAG_HALF=1) against the oracle on C2-shaped batches of rising divergence:
distances within and beyond its threshold of 48 (dev tool)."""
import os, sys
os.environ["MALCS_DIAG_HALF"] = "1"
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import gen
from algebra_amd import _native as N
from oracle import oracle as O
ctx = N.Context(0)
bad_total = 0
for pairs, length, p, seed in [(20000, 1000, 0.01, 5), (20000, 1000, 0.02, 6), (8000, 1000, 0.03, 7), (20000, 2000, 0.005, 8), (6000, 700, 0.0, 9)]:
    data = gen.c2_batch(pairs, seed=seed, length=length, p=p)
    b = N.Batch(ctx, *data, 4)
    d = b.distances()
    d2 = b.distances()
    ref = O.edit_distance_batch(*data)
    bad = int((d != ref).sum()) + int((d2 != ref).sum())
    bad_total += bad
    print(f"pairs={pairs} length={length} p={p}: distances {int(ref.min())}..{int(ref.max())}, beyond 48: {int((ref > 48).sum())}, mismatches={bad}", flush=True)
    b.close()
sys.exit(1 if bad_total else 0)
