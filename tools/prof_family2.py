"""Family 2's bit-vector kernels at the bench's size (extra.c2_lcs_graphs_bitvector: 100 000 C2 pairs, sweeps off), for the
profile passes of tools/prof_family2.sh."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gen
from algebra_amd import _native as N
os.environ["MALCS_SWEEP_DMAX"] = "-1"
os.environ.setdefault("MALCS_GRAPH_RESIDENT_BAND", "0")     # every group's tie bits stored: the HBM-bound case the committed profile is of (1: the band only, the default of the library)
ctx = N.Context(0)
b = N.Batch(ctx, *gen.c2_batch(100000, seed=12), 4)
for _ in range(int(os.environ.get("REPS", "3"))):
    g = b.graphs()
    print("tie bytes", g.tie_bytes, "pairs", g.tie_pairs, "forward ms %.3f reach ms %.3f" % (g.timing[0], g.timing[1]), flush=True)
    g.close()
b.close()
