"""Where the host time of compare_matrix goes (cProfile over ten matrices of BASELINE config 4; dev tool)."""
import cProfile, pstats, sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import gen
from algebra_amd.lcs import LCSgraph
from algebra_amd.relations import compare_matrix
from algebra_amd.variants import Variant

case = gen.matrix_case(seed=4, length=100000, n_clusters=50, per_cluster=100, width=200)
ref = case["reference"]
graphs = LCSgraph.from_variant_batch(ref, [[Variant(*v)] for v in case["variants"]])
print("supremal lengths:", sum(len(g.supremal.sequence) for g in graphs), "observed:", sum(len(g._observed) for g in graphs),
      "handles:", len({id(g._result[0]) for g in graphs}))
compare_matrix(ref, graphs)
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    compare_matrix(ref, graphs)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(25)
