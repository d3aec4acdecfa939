"""The bit-vector graph kernels (k_lcs_forward / k_lcs_reach) on C2-shaped batches of several sizes: stage times by HIP
events and the rate the tie bits move at (0.25 B per cell each way; dev tool)."""
import os, sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
os.environ["MALCS_SWEEP_DMAX"] = "-1"
import gen
from algebra_amd import _native as N
ctx = N.Context(0)
for n in [int(x) for x in (sys.argv[1:] or ["20000", "50000", "100000"])]:
    b = N.Batch(ctx, *gen.c2_batch(n, seed=12), 4)
    b.graphs().close()
    for _ in range(2):
        t0 = time.perf_counter(); g = b.graphs(); wall = time.perf_counter() - t0
        f, r, h = (float(x) for x in g.timing)
        print("n=%d wall %.1f ms forward %.2f ms (%.0f GB/s) reach %.2f ms (%.0f GB/s) host build %.1f ms" %
              (n, wall * 1e3, f, n * 1e6 * 0.25 / f / 1e6, r, n * 1e6 * 0.25 / r / 1e6, h), flush=True)
        g.close()
    b.close()
