"""Dev tool, with tools/plan_clocks.sh: per-workgroup and per-phase times inside k_plan_ends (C2 batch, last of five plans)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, gen
from algebra_amd import _native as N
ctx = N.Context(0)
n = 100000
b = N.Batch(ctx, *gen.c2_batch(n, seed=2), 4)
for _ in range(5):
    b.replan(); b.launch_distances()
ctx.sync()
t = b.plan_dump(9).astype(np.int64)
forced = bool(os.environ.get("MALCS_DIAG_HALF"))
stride = n // 1024
n_sample = (n - stride // 2 + stride - 1) // stride
parts = int(os.environ.get("PARTS", "8"))                       # lanes per sampled pair (1 kb pairs in parts of 128 symbols)
nsb = 0 if forced else -(-n_sample // (256 // parts)) if not os.environ.get("MALCS_SAMPLE_WAVE") else 16      # sample workgroups
nwg = nsb + (n + 255) // 256
st, en = t[0:2 * nwg:2], t[1:2 * nwg:2]
t0 = st.min()
st = (st - t0) * 10; en = (en - t0) * 10                       # ns
if nsb:
    print("sample workgroups: start", st[:nsb].min(), st[:nsb].max(), "end", en[:nsb].min(), en[:nsb].max())
print("ends workgroups: start min/med/max", st[nsb:].min(), int(np.median(st[nsb:])), st[nsb:].max(), "end min/med/p90/max",
      en[nsb:].min(), int(np.median(en[nsb:])), int(np.percentile(en[nsb:], 90)), en[nsb:].max())
print("last to end:", [(int(i), int(st[i]), int(en[i])) for i in np.argsort(en)[-6:]])
if nsb and not os.environ.get("MALCS_SAMPLE_WAVE"):
    ph = (t[1200:1205] - t[1200]) * 10
    print("sample workgroup 0, ns since its start: table", ph[1], "sequences in LDS", ph[2], "cuts", ph[3], "lanes", ph[4], "end", (t[1] - t[1200]) * 10)
b.close()
