"""Saves what the device sweeps hand to the host builder for the C3 pairs (dev tool): gpurun_out/c3_runs.npz."""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import gen
from algebra_amd import _native as N
from algebra_amd.lcs.encoding import encode_pairs

pairs = [gen.tandem_pair(np.random.RandomState(300 + s), 10000, 20) for s in range(16)]
syms, ao, al, bo, bl, ns = encode_pairs(pairs)
bt = N.Batch(N.default_context(), syms, ao, al, bo, bl, ns)
g = bt.graphs()
print(g.counts[:4], g.timing)
runs = [g.runs(i).astype(np.int32) for i in range(16)]
np.savez_compressed("gpurun_out/c3_runs.npz", counts=g.counts, first=np.concatenate([[0], np.cumsum([len(r) for r in runs])]),
                    runs=np.concatenate(runs), m=al, n=bl)
