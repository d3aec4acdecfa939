"""cProfile of single facade calls (dev tool)."""
import sys, cProfile, pstats
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import gen
from algebra_amd.lcs import LCSgraph
rs = np.random.RandomState(1)
a = gen.random_seq(rs, 100); b = gen.mutate(rs, a, 0.02, 0.02)
LCSgraph(a, b)
pr = cProfile.Profile(); pr.enable()
for _ in range(300): LCSgraph(a, b)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
