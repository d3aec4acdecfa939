"""cProfile of one compare_matrix call on the C4 case (dev tool)."""
import cProfile, pstats, sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import gen
from algebra_amd.lcs import LCSgraph
from algebra_amd.relations import compare_matrix
from algebra_amd.variants import Variant
case = gen.matrix_case(seed=4, length=100000, n_clusters=50, per_cluster=100, width=200)
graphs = LCSgraph.from_variant_batch(case["reference"], [[Variant(*v)] for v in case["variants"]])
compare_matrix(case["reference"], graphs)
stats = {}
t0 = time.perf_counter(); compare_matrix(case["reference"], graphs, stats=stats); print("wall", time.perf_counter() - t0, stats)
pr = cProfile.Profile(); pr.enable(); compare_matrix(case["reference"], graphs); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
