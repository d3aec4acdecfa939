"""BASELINE config 4 on one GPU: all-vs-all relation matrix over clustered variants (dev tool)."""
import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import gen
from algebra_amd.lcs import LCSgraph
from algebra_amd.relations import compare_matrix
from algebra_amd.variants import Variant

n_clusters, per = int(sys.argv[1]) if len(sys.argv) > 1 else 50, int(sys.argv[2]) if len(sys.argv) > 2 else 100
case = gen.matrix_case(seed=4, length=100000, n_clusters=n_clusters, per_cluster=per, width=200)
ref = case["reference"]
alleles = [[Variant(*v)] for v in case["variants"]]
t0 = time.time()
graphs = LCSgraph.from_variant_batch(ref, alleles)
t1 = time.time()
print(f"supremals: {len(graphs)} alleles in {t1-t0:.2f}s", flush=True)
compare_matrix(ref, graphs)                      # warm-up: code objects, atom masks
t1 = time.time()
stats = {}
labels = compare_matrix(ref, graphs, stats=stats)
t2 = time.time()
print({k: round(v * 1e3, 1) for k, v in stats.items() if k.startswith("seconds_")}, "ms")
n = len(graphs)
pairs = n * (n - 1) // 2
print(f"matrix: {n}x{n} ({pairs} pairs) in {t2-t1:.2f}s -> {pairs/(t2-t1):.3g} relations/s; dp_pairs={stats['dp_pairs']} "
      f"labels={np.bincount(labels.reshape(-1), minlength=5).tolist()}", flush=True)
