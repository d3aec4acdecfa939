import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
from algebra_amd import _native as N
from algebra_amd.lcs.encoding import encode_pairs
from oracle import oracle as O
import gen
ctx = N.Context(0)
rs = np.random.RandomState(3)
for m in (50, 100, 200, 255, 256, 300, 600, 1000, 2000, 4000, 4100, 5000, 10000):
    for p in (0.01,):
        a = gen.random_seq(rs, m); b = gen.mutate(rs, a, p, p)
        syms, ao, al, bo, bl, ns = encode_pairs([(a, b)])
        bt = N.Batch(ctx, syms, ao, al, bo, bl, ns)
        try:
            g = bt.graphs()
            ref = O.lcs_graph(a, b)
            ok = (g.distance[0] == ref.distance and tuple(g.supremal[0]) == ref.supremal and np.array_equal(g.nodes(0), ref.nodes) and np.array_equal(g.edges(0), ref.edges))
            print(m, "ok" if ok else "MISMATCH", g.distance[0], ref.distance, len(g.runs(0)), flush=True)
        except N.MalcsError as e:
            print(m, "ERROR", str(e)[:100], flush=True)
        bt.close()
