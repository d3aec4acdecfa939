"""Where the host time of LCSgraph.from_variant_batch goes for BASELINE config 4's 5 000 variants (cProfile; dev tool)."""
import cProfile, pstats, sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import gen
from algebra_amd.lcs import LCSgraph
from algebra_amd.variants import Variant

case = gen.matrix_case(seed=4, length=100000, n_clusters=50, per_cluster=100, width=200)
ref = case["reference"]
alleles = [[Variant(*v)] for v in case["variants"]]
LCSgraph.from_variant_batch(ref, alleles)
for _ in range(3):
    t0 = time.perf_counter(); LCSgraph.from_variant_batch(ref, alleles); print("%.1f ms" % ((time.perf_counter() - t0) * 1e3))
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    LCSgraph.from_variant_batch(ref, alleles)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
