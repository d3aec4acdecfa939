"""Latency of single calls through the facade (dev tool): what a caller of the reference's API sees per call."""
import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import gen
import algebra_amd
from algebra_amd.lcs import LCSgraph
from algebra_amd.relations import compare
from algebra_amd.variants import Variant
rs = np.random.RandomState(1)
for m in (20, 100, 1000):
    a = gen.random_seq(rs, m); b = gen.mutate(rs, a, 0.02, 0.02)
    algebra_amd.edit_distance(a, b); LCSgraph(a, b)
    t0 = time.perf_counter()
    for _ in range(200): algebra_amd.edit_distance(a, b)
    t1 = time.perf_counter()
    for _ in range(200): LCSgraph(a, b)
    t2 = time.perf_counter()
    for _ in range(200): LCSgraph.from_sequence(a, b)
    t3 = time.perf_counter()
    print(f"m={m}: edit_distance {1e6*(t1-t0)/200:.0f} us, LCSgraph {1e6*(t2-t1)/200:.0f} us, from_sequence {1e6*(t3-t2)/200:.0f} us per call", flush=True)
ref = gen.random_seq(rs, 1000)
t0 = time.perf_counter()
for i in range(200): compare(ref, [Variant(100 + i, 101 + i, "T")], [Variant(100 + i, 102 + i, "")])
print(f"compare(): {1e6*(time.perf_counter()-t0)/200:.0f} us per call")
