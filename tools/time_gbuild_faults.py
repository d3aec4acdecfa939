"""Dev tool: wall time and minor page faults of consecutive Batch.graphs() calls over C2's 100 000 pairs (a call whose
big host arrays come back from the allocator unmapped pays 1-2 ms of faults)."""
import os, sys, time, resource
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gen
from algebra_amd import _native as N
ctx = N.Context(0)
b = N.Batch(ctx, *gen.c2_batch(100000, seed=12), 4)
for k in range(12):
    f0 = resource.getrusage(resource.RUSAGE_SELF).ru_minflt
    t0 = time.perf_counter(); g = b.graphs(); wall = (time.perf_counter() - t0) * 1e3
    f1 = resource.getrusage(resource.RUSAGE_SELF).ru_minflt
    t0 = time.perf_counter(); g.close(); cl = (time.perf_counter() - t0) * 1e3
    f2 = resource.getrusage(resource.RUSAGE_SELF).ru_minflt
    print("call", k, "wall_ms %.2f" % wall, "minor faults in call", f1 - f0, "close_ms %.2f" % cl, "faults in close", f2 - f1, flush=True)
b.close()
