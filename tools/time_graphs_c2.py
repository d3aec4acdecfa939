"""Wall time of LCS graphs for a C2-shaped batch of 100 000 pairs next to the library's own split (dev tool;
MALCS_TRACE=1 adds the host build's thread totals)."""
import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import gen
from algebra_amd import _native as N
ctx = N.Context(0)
b = N.Batch(ctx, *gen.c2_batch(100000, seed=12), 4)
b.graphs().close()
for _ in range(3):
    t0 = time.perf_counter(); g = b.graphs(); t1 = time.perf_counter()
    print("wall %.1f ms, timing %s, %d built on the device in %.2f ms" % ((t1 - t0) * 1e3, [round(float(x), 2) for x in g.timing], g.device_built, g.device_build_ms), flush=True)
    t2 = time.perf_counter(); g.close(); print("close %.1f ms" % ((time.perf_counter() - t2) * 1e3))
