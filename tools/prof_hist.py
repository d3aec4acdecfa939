"""Histogram of a kernel's durations from a rocprofv3 --kernel-trace result directory (dev tool):
    python tools/prof_hist.py DIR [name-pattern] [bin-us]"""
import glob, sqlite3, sys
import numpy as np
db = sqlite3.connect(glob.glob(sys.argv[1] + "/**/*.db", recursive=True)[0])
like = sys.argv[2] if len(sys.argv) > 2 else "%k_llcs_diag%"
step = float(sys.argv[3]) if len(sys.argv) > 3 else 0.25
rows = db.execute("select start, end from kernels where name like ? order by start", (like,)).fetchall()
a = np.array(rows, dtype=np.int64)
d = (a[:, 1] - a[:, 0]) / 1e3
gap = (a[1:, 0] - a[:-1, 1]) / 1e3
print(f"{len(d)} launches: mean {d.mean():.2f} median {np.median(d):.2f} min {d.min():.2f} p10 {np.percentile(d,10):.2f} p90 {np.percentile(d,90):.2f} max {d.max():.2f} us")
print(f"gap to the next launch of the same kernel: median {np.median(gap):.2f} us")
lo = np.floor(d.min() / step) * step
for b in np.arange(lo, np.percentile(d, 99) + step, step):
    n = int(((d >= b) & (d < b + step)).sum())
    print(f"  {b:6.2f} {n:5d} {'#' * (n * 200 // len(d))}")
# in time order: runs of fast / slow launches?
print("first 60 in order:", " ".join(f"{x:.1f}" for x in d[200:260]))
