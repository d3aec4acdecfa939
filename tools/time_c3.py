"""C3 (16 x 10 kb tandem-repeat windows) graphs, repeated: wall and the library's own split (dev tool)."""
import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import gen
from algebra_amd import _native as N
from algebra_amd.lcs.encoding import encode_pairs
pairs = [gen.tandem_pair(np.random.RandomState(300 + s), 10000, 20) for s in range(16)]
syms, ao, al, bo, bl, ns = encode_pairs(pairs)
bt = N.Batch(N.default_context(), syms, ao, al, bo, bl, ns)
host, wall = [], []
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 12):
    t0 = time.perf_counter(); g = bt.graphs(); wall.append((time.perf_counter() - t0) * 1e3); host.append(float(g.timing[2])); g.close()
print("host build ms:", " ".join(f"{x:.1f}" for x in host), "| median of the last 10: %.2f" % np.median(host[-10:]))
print("wall ms:", " ".join(f"{x:.1f}" for x in wall), "| median of the last 10: %.2f" % np.median(wall[-10:]))
