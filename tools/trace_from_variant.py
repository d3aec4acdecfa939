"""Dev tool: stages of LCSgraph.from_variant_batch on the C4 case (MALCS_TRACE=1 on the third call)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gen
from algebra_amd import _native as N
from algebra_amd.lcs import LCSgraph
from algebra_amd.variants import Variant
ctx = N.Context(0)
case = gen.matrix_case(seed=4, length=100000, n_clusters=50, per_cluster=100, width=200)
variants = [[Variant(*v)] for v in case["variants"]]
for _ in range(2):
    LCSgraph.from_variant_batch(case["reference"], variants, ctx=ctx)
os.environ["MALCS_TRACE"] = "1"
t0 = time.perf_counter()
g = LCSgraph.from_variant_batch(case["reference"], variants, ctx=ctx)
print("wall ms", (time.perf_counter() - t0) * 1e3)
del os.environ["MALCS_TRACE"]
for _ in range(3):
    t0 = time.perf_counter()
    g = LCSgraph.from_variant_batch(case["reference"], variants, ctx=ctx)
    print("wall ms", (time.perf_counter() - t0) * 1e3)
