"""Dev tool: plan + launch per batch with two batches in flight (two contexts = two streams), against one:
    python tools/time_two_in_flight.py [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, gen
from algebra_amd import _native as N
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
NF = int(os.environ.get('NF', '3'))
ctxs = [N.Context(0) for _ in range(NF)]
bs = [N.Batch(c, *gen.c2_batch(100000, seed=2 + i), 4) for i, c in enumerate(ctxs)]
for b in bs:
    b.distances()
for k in range(1, NF + 1):
    out = []
    for rep in range(4):
        for i in range(60):
            b = bs[i % k]; b.replan(); b.launch_distances()
        for c in ctxs: c.sync()
        t0 = time.perf_counter()
        for i in range(steps):
            b = bs[i % k]; b.replan(); b.launch_distances()
        enq = (time.perf_counter() - t0) / steps * 1e6
        for c in ctxs: c.sync()
        out.append((time.perf_counter() - t0) / steps * 1e6)
        out.append(enq)
    print(k, "in flight: us per batch (total, host enqueue alone) x4:", " ".join("%.2f" % x for x in out))
for b in bs: b.close()
