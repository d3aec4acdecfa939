"""Dev tool: the window kernel's per-wave time stamps (MALCS_DIAG_STAMPS) right behind a plan against a replay over
the rows of a plan that is already there:  python tools/diag_in_step.py [pairs]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import gen
from algebra_amd import _native as N

pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
ctx = N.Context(0)
b = N.Batch(ctx, *gen.c2_batch(pairs, seed=2), 4)
b.distances()
path = "/tmp/diag_stamps.bin"


def capture(in_step):
    for _ in range(20):
        b.replan(); b.launch_distances()
    ctx.sync()
    os.environ["MALCS_DIAG_STAMPS"] = path
    for _ in range(3):                     # the first makes the buffer; the last is read
        if in_step:
            b.replan()
        b.launch_distances()
        ctx.sync()
    del os.environ["MALCS_DIAG_STAMPS"]
    s = np.fromfile(path, dtype=np.uint64).reshape(-1, 6).astype(np.int64)
    return s[s[:, 0] > 0]


for name, in_step in (("replayed", False), ("in a step", True), ("replayed", False), ("in a step", True)):
    s = capture(in_step)
    t = (s[:, :4] - s[:, 0].min()) * 0.01                      # us
    blocks = ((s[:, 5] >> 8) & ((1 << 28) - 1)).astype(float)
    first = np.argsort(s[:, 0])[:1024]                        # the waves that start a SIMD's work
    rest = np.argsort(s[:, 0])[1024:]
    print(f"{name:10s}: {len(s)} waves, start median {np.median(t[:,0]):.2f} p90 {np.percentile(t[:,0], 90):.2f} max {t[:,0].max():.2f}; "
          f"setup median {np.median(t[:,1]-t[:,0]):.2f} (first 1024: {np.median((t[:,1]-t[:,0])[first]):.2f}, later: {np.median((t[:,1]-t[:,0])[rest]):.2f}); "
          f"loop per block first {np.median(((t[:,2]-t[:,1])/blocks)[first]):.3f} later {np.median(((t[:,2]-t[:,1])/blocks)[rest]):.3f}; "
          f"join median {np.median(t[:,3]-t[:,2]):.2f}; loop end max {t[:,2].max():.2f}; done median {np.median(t[:,3]):.2f} max {t[:,3].max():.2f}")
    xcc = s[:, 5] & 15
    hw = s[:, 4]
    simd_key = (xcc << 20) | (hw & 0xfff30)                        # xcc, se / sh / cu, simd
    wg = np.arange(len(s))
    keys, inv, cnt = np.unique(simd_key, return_inverse=True, return_counts=True)
    tot = np.bincount(inv, weights=blocks)
    same = sum(int(simd_key[i] == simd_key[i + 1024]) for i in range(len(s) - 1024))
    print(f"      {len(keys)} SIMDs used; waves per SIMD: {dict(zip(*np.unique(cnt, return_counts=True)))}; blocks per SIMD min {tot.min():.0f} median {np.median(tot):.0f} max {tot.max():.0f}; "
          f"workgroups w and w + 1024 on one SIMD: {same} of {len(s) - 1024}")
    for k in list(np.argsort(-tot)[:3]) + [0, 1]:
        print(f"      SIMD {int(keys[k]):#x}: workgroups {wg[inv == k].tolist()} blocks {blocks[inv == k].astype(int).tolist()}")
    late = np.argsort(-t[:, 3])[:4]
    flags = (s[:, 5] >> 36) & 0xf                                  # PIECE_* flags (bits 28-31 of piece_info)
    for w in late:
        print(f"      late: blocks {int(blocks[w])} flags {int(flags[w]):#x} start {t[w,0]:.2f} setup {t[w,1]-t[w,0]:.2f} loop {t[w,2]-t[w,1]:.2f} ({(t[w,2]-t[w,1])/blocks[w]:.3f} / block) join {t[w,3]-t[w,2]:.2f} done {t[w,3]:.2f} xcc {int(s[w,5] & 15)} hw {int(s[w,4]):#x}")
    per = (t[:, 2] - t[:, 1]) / blocks
    print(f"      loop per block: p50 {np.percentile(per, 50):.3f} p90 {np.percentile(per, 90):.3f} p99 {np.percentile(per, 99):.3f} max {per.max():.3f}; setup p50 {np.percentile(t[:,1]-t[:,0], 50):.2f} p90 {np.percentile(t[:,1]-t[:,0], 90):.2f} p99 {np.percentile(t[:,1]-t[:,0], 99):.2f} max {(t[:,1]-t[:,0]).max():.2f}; join p99 {np.percentile(t[:,3]-t[:,2], 99):.2f} max {(t[:,3]-t[:,2]).max():.2f}")
b.close()
