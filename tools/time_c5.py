"""Time BASELINE config 5 shapes: 1 Mb pairs, (i) ~20 small indels, (ii) 2 % indels (dev tool)."""
import os, sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import gen
from algebra_amd import _native as N
from algebra_amd.lcs.encoding import encode_pairs
ctx = N.Context(0)
rs = np.random.RandomState(5)
which = sys.argv[1:] or ["i", "ii"]
for tag in which:
    pairs = []
    for j in range(4 if tag == "i" else 1):
        a = gen.random_seq(rs, 1000000)
        if tag == "i":
            b = list(a)
            for _ in range(20):
                p = rs.randint(0, len(b)); l = rs.randint(1, 4)
                if rs.randint(0, 2): del b[p:p + l]
                else: b[p:p] = list(gen.random_seq(rs, l))
            b = "".join(b)
        else:
            b = gen.mutate(rs, a, 0.01, 0.01)
        pairs.append((a, b))
    data = encode_pairs(pairs)
    t0 = time.time(); bt = N.Batch(ctx, *data); t1 = time.time()
    d = bt.distances(); t2 = time.time()
    d = bt.distances(); t3 = time.time()
    g = bt.graphs(); t4 = time.time()
    tm = g.timing
    print(f"C5({tag}) pairs={len(pairs)} D={d.tolist()} batch {t1-t0:.3f}s dist {t3-t2:.4f}s graphs {t4-t3:.3f}s "
          f"(pass1 {tm[0]:.1f} ms, pass2 {tm[1]:.1f} ms, host {tm[2]:.1f} ms) nodes={g.counts[:,2].tolist()}", flush=True)
    g.close(); bt.close()
