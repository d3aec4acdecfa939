"""Timeline of one launch of the window kernel from per-wave time stamps (dev tool; MALCS_DIAG_STAMPS):
    python tools/diag_timeline.py [pairs]"""
import os, sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import gen
from algebra_amd import _native as N

pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
ctx = N.Context(0)
b = N.Batch(ctx, *gen.c2_batch(pairs, seed=2), 4)
b.distances()
for _ in range(20): b.time_distances(50)          # warm clocks
path = "/tmp/diag_stamps.bin"
os.environ["MALCS_DIAG_STAMPS"] = path
b.time_dominant_kernel(1)
del os.environ["MALCS_DIAG_STAMPS"]
s = np.fromfile(path, dtype=np.uint64).reshape(-1, 6).astype(np.int64)
t0 = s[:, 0].min()
t = (s[:, :4] - t0) * 10.0 / 1000.0                 # us (100 MHz clock)
n = len(s)
print(f"{n} waves; start: min {t[:,0].min():.2f} median {np.median(t[:,0]):.2f} max {t[:,0].max():.2f} us")
print(f"setup (start -> loop): median {np.median(t[:,1]-t[:,0]):.2f} max {(t[:,1]-t[:,0]).max():.2f}")
print(f"loop: median {np.median(t[:,2]-t[:,1]):.2f} min {(t[:,2]-t[:,1]).min():.2f} max {(t[:,2]-t[:,1]).max():.2f}")
print(f"loop end: min {t[:,2].min():.2f} median {np.median(t[:,2]):.2f} max {t[:,2].max():.2f}")
print(f"join (loop end -> done): median {np.median(t[:,3]-t[:,2]):.2f} max {(t[:,3]-t[:,2]).max():.2f}")
print(f"done: median {np.median(t[:,3]):.2f} max {t[:,3].max():.2f}")
order = np.argsort(s[:, 0])
for q in (0, n // 8, n // 4, n // 2, 3 * n // 4, n - 1):
    w = order[q]
    print(f"  wave started {q:5d}th: wg {w:5d} start {t[w,0]:.2f} loop {t[w,1]:.2f} -> {t[w,2]:.2f} done {t[w,3]:.2f} hw {s[w,4]:#x} xcc {s[w,5] & 15}")
# start time against workgroup number
for lo in range(0, n, max(1, n // 12)):
    hi = min(n, lo + max(1, n // 12))
    print(f"  wg {lo:5d}-{hi:5d}: start {t[lo:hi,0].mean():.2f} loop-end {t[lo:hi,2].mean():.2f} done {t[lo:hi,3].mean():.2f}")

# per SIMD (workgroups w, w + 1024, ... share one): pieces, blocks, when the last one is done
S = 1024
blocks = (s[:, 5] >> 8) & ((1 << 28) - 1)
simd = np.arange(n) % S
cnt = np.bincount(simd, minlength=S); tot = np.bincount(simd, weights=blocks, minlength=S)
last = np.zeros(S); np.maximum.at(last, simd, t[:, 3])
lend = np.zeros(S); np.maximum.at(lend, simd, t[:, 2])
for c in np.unique(cnt):
    for bsum in np.unique(tot[cnt == c]):
        sel = (cnt == c) & (tot == bsum)
        print(f"  SIMDs with {c} pieces, {int(bsum)} blocks: {sel.sum():4d}; loop end {lend[sel].mean():.2f} (max {lend[sel].max():.2f}), done {last[sel].mean():.2f} (max {last[sel].max():.2f})")
late = np.argsort(-t[:, 3])[:8]
for w in late:
    print(f"  late: wg {w} simd {w % S} blocks {blocks[w]} start {t[w,0]:.2f} loop {t[w,1]:.2f} -> {t[w,2]:.2f} done {t[w,3]:.2f} xcc {s[w,5] & 15} hw {s[w,4]:#x}")

xcc = s[:, 5] & 15
for x in range(8):
    sel = xcc == x
    print(f"  xcc {x}: {sel.sum()} waves, blocks {blocks[sel].sum()}, setup {np.median(t[sel,1]-t[sel,0]):.2f}, loop per block {np.median((t[sel,2]-t[sel,1])/blocks[sel]):.3f}, loop end median {np.median(t[sel,2]):.2f} max {t[sel,2].max():.2f}, done max {t[sel,3].max():.2f}")

# per CU: waves, blocks, when its last loop ends
hw = s[:, 4]
cu_key = (xcc << 16) | (hw & 0xff00)
simd_id = (hw >> 4) & 3
rows_ = []
for key in np.unique(cu_key):
    sel = cu_key == key
    per_simd = [int(blocks[sel & (simd_id == q)].sum()) for q in range(4)]
    rows_.append((int(sel.sum()), int(blocks[sel].sum()), max(per_simd), float(t[sel, 2].max()), float(np.median(t[sel, 2])), float(t[sel, 3].max())))
rows_ = np.array(rows_)
print(f"{len(rows_)} CUs")
for w in np.unique(rows_[:, 0]):
    sel = rows_[:, 0] == w
    print(f"  CUs with {int(w)} waves: {sel.sum()}, blocks {rows_[sel,1].mean():.1f}, busiest SIMD {rows_[sel,2].mean():.1f} blocks, last loop end {rows_[sel,3].mean():.2f} (max {rows_[sel,3].max():.2f}), median loop end {rows_[sel,4].mean():.2f}")
for b in np.unique(rows_[:, 1]):
    sel = rows_[:, 1] == b
    print(f"  CUs with {int(b)} blocks: {sel.sum()}, last loop end {rows_[sel,3].mean():.2f} (max {rows_[sel,3].max():.2f})")
# do the workgroups w, w + 1024, ... really share a SIMD?
same = 0
for i in range(min(S, n - S)):
    same += int(cu_key[i] == cu_key[i + S] and simd_id[i] == simd_id[i + S])
print(f"workgroups w and w + {S} on the same SIMD: {same} of {min(S, n - S)}")
