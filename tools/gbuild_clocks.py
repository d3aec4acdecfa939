import os, sys, ctypes, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gen
from algebra_amd import _native as N
lib = ctypes.CDLL(os.environ["MALCS_LIB"])
ctx = N.Context(0)
b = N.Batch(ctx, *gen.c2_batch(100000, seed=12), 4)
b.graphs().close()
g = b.graphs(); ms = float(g.device_build_ms); g.close()
o = np.zeros(5 * 131072, np.uint32)
lib.malcs_dev_gb_clocks(o.ctypes.data_as(ctypes.c_void_p))
o = o.reshape(-1, 5)[:100000].astype(np.float64)
print("device_build_ms", ms, "mean us per pair: build %.1f count_reachable %.1f cursor %.1f write %.1f" % tuple(o[:, :4].mean(0) / 100))
print("median: build %.1f count %.1f cursor %.1f write %.1f" % tuple(np.median(o[:, :4], 0) / 100))
