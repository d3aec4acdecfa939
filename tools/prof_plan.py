"""Kernel trace of the window plan on resident planes (dev tool): rocprofv3 --kernel-trace --stats -- python tools/prof_plan.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gen
from algebra_amd import _native as N
ctx = N.Context(0)
b = N.Batch(ctx, *gen.c2_batch(100000, seed=2), 4)
for _ in range(20):
    b.replan()
    b.launch_distances()
ctx.sync()
b.close()
