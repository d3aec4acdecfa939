"""Host half of the graph path on C3 input, no device: the level lists the device sweeps produce for the 16 x 10 kb
tandem-repeat pairs (saved by tools/dump_c3_runs.py on the GPU box) through malcs_graphs_from_runs (dev tool)."""
import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
from algebra_amd import _native as N

z = np.load(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/c3_runs.npz")
counts, first, runs, m, n = z["counts"], z["first"], z["runs"].astype(np.int64), z["m"], z["n"]
pairs = len(m)
for sel in ([0], [3], list(range(pairs))):
    f = np.concatenate([[0], np.cumsum([first[i + 1] - first[i] for i in sel])])
    flat = np.concatenate([runs[first[i]:first[i + 1]] for i in sel])
    for rep in range(5):
        t0 = time.perf_counter()
        res = N.graphs_from_runs(m[sel], n[sel], [0] * len(sel), counts[sel, 1], flat, f)
        t1 = time.perf_counter()
        ok = np.array_equal(res.counts[:, 2:], counts[sel, 2:])
        res.close()
        print(f"graphs {sel if len(sel) < 3 else len(sel)}: {1e3 * (t1 - t0):.1f} ms, edges {int(counts[sel, 3].sum())}, counts equal: {ok}", flush=True)
