"""Dev tool: C2's 100 000 graphs from NT host threads, a context and a batch each (ctypes releases the GIL for the call):
one thread's host loops run under the other's kernels.  Prints graphs/s over all threads."""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gen
from algebra_amd import _native as N
NT = int(os.environ.get("NT", "2")); REPS = int(os.environ.get("REPS", "6"))
ctxs = [N.Context(0) for _ in range(NT)]
batches = [N.Batch(ctxs[t], *gen.c2_batch(100000, seed=12 + t), 4) for t in range(NT)]
for b in batches:
    for _ in range(2): b.graphs().close()
def work(t):
    for _ in range(REPS):
        g = batches[t].graphs(); assert int(g.device_built) == 100000; g.close()
t0 = time.perf_counter()
th = [threading.Thread(target=work, args=(t,)) for t in range(NT)]
for x in th: x.start()
for x in th: x.join()
dt = time.perf_counter() - t0
print("threads", NT, "calls", NT * REPS, "ms per 100 000 graphs %.2f" % (dt * 1e3 / (NT * REPS)), "graphs/s %.3g" % (NT * REPS * 100000 / dt))
for b in batches: b.close()
