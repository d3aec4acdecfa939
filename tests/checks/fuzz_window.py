"""Window class of family 1 against the oracle on pairs of mixed lengths, length differences and indel rates, so that
tiles mix lanes whose plain stretches (malcs_diag.hip: three-block statement) start and end in different blocks
(dev tool).  python tests/checks/fuzz_window.py [pairs] [seed] [symbols]"""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import gen
from algebra_amd import _native as N
from oracle import oracle as O

n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
nsym = int(sys.argv[3]) if len(sys.argv) > 3 else 4
data = gen.window_mix(n_pairs, int(sys.argv[2]) if len(sys.argv) > 2 else 7, nsym)
lens = np.stack([data[2], data[4]], 1)
ctx = N.Context(0)
b = N.Batch(ctx, *data, nsym)
info = b.plan_info()
d = b.distances()
ref = O.edit_distance_batch(*data)
bad = np.nonzero(d != ref)[0]
print(f"pairs={n_pairs} window_pairs={info['window_pairs']} full_classes={info['full_classes']} "
      f"distances 0..{int(ref.max())} beyond_window={(ref > info['window_threshold']).sum()} mismatches={len(bad)}", flush=True)
for i in bad[:8]:
    print("   pair", i, "lengths", lens[i].tolist(), "gpu", d[i], "oracle", ref[i])
b.close()
sys.exit(1 if len(bad) else 0)
