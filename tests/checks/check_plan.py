"""The window class's plan, as the device made it, against what it must satisfy (dev tool; tests/test_gpu_lcs.py has the
same checks as a test): python tests/checks/check_plan.py [pairs] [length] [p] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import gen
from algebra_amd import _native as N
from util import check_window_plan

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
length = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
p = float(sys.argv[3]) if len(sys.argv) > 3 else 0.01
seed = int(sys.argv[4]) if len(sys.argv) > 4 else 2
data = gen.c2_batch(n, seed=seed, length=length, p=p)
b = N.Batch(N.default_context(), *data, 4)
print(check_window_plan(b))
b.close()
from oracle import oracle as O
b = N.Batch(N.default_context(), *data, 4)
d0 = b.distances()
ref = O.edit_distance_batch(*data)
print("first distances ok:", np.array_equal(d0, ref))
for k in range(6):
    b.replan()
    try:
        print("replan", k, check_window_plan(b))
    except AssertionError as e:
        print("replan", k, "PLAN BROKEN:", repr(e))
    b.launch_distances()
    d = b.fetch_distances()
    print("   distances ok:", np.array_equal(d, ref), int((d != ref).sum()))
b.close()
b2 = N.Batch(N.default_context(), data[0], data[3], data[4], data[1], data[2], 4)
try:
    print("second batch", check_window_plan(b2))
except AssertionError as e:
    print("second batch PLAN BROKEN:", repr(e))
print("   distances ok:", np.array_equal(b2.distances(), ref))
b2.close()
