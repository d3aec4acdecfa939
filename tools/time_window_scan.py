"""Window kernel time against sequence length and wave count (dev tool): fixed cost per launch and the
step from one to two waves per SIMD.  python tools/time_window_scan.py"""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import gen
from algebra_amd import _native as N

ctx = N.Context(0)
for pairs in (32768, 65536, 100000, 131072, 196608):
    for length in (125, 250, 500, 1000, 2000):
        b = N.Batch(ctx, *gen.c2_batch(pairs, seed=2, length=length), 4)
        info = b.plan_info()
        b.distances()
        b.time_dominant_kernel(5)
        k = b.time_dominant_kernel(50) / 50
        print(f"pairs={pairs} waves={(pairs + 63) // 64} length={length} window_pairs={info['window_pairs']} kernel={k * 1e3:.1f} us", flush=True)
        b.close()
