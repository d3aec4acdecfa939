"""Quick GPU check of family 1 against the oracle on random pairs (dev tool)."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from algebra_amd import _native as N
from oracle import oracle as O

rng = np.random.default_rng(1)
def mutate(a, p):
    out = []
    for ch in a:
        r = rng.random()
        if r < p: continue
        if r < 2 * p: out.append(rng.integers(0, 4))
        out.append(ch)
    return np.array(out, dtype=np.uint8)

def make(lengths, p, nsym=4):
    seqs = []
    for L in lengths:
        a = rng.integers(0, nsym, L).astype(np.uint8)
        b = mutate(a, p) if rng.random() < 0.8 else rng.integers(0, nsym, max(0, L + rng.integers(-5, 6))).astype(np.uint8)
        seqs.append((a, b))
    syms = np.concatenate([x for ab in seqs for x in ab] + [np.zeros(1, np.uint8)])
    a_off, a_len, b_off, b_len = [], [], [], []
    pos = 0
    for a, b in seqs:
        a_off.append(pos); a_len.append(len(a)); pos += len(a)
        b_off.append(pos); b_len.append(len(b)); pos += len(b)
    return syms, np.array(a_off, np.uint64), np.array(a_len, np.uint32), np.array(b_off, np.uint64), np.array(b_len, np.uint32)

ctx = N.Context(0)
print(ctx.device_info())
ok = True
for name, lengths, nsym in [
    ("tiny", list(range(0, 70)) * 2, 4),
    ("word-boundaries", [63, 64, 65, 127, 128, 129, 255, 256, 257, 511, 512, 513, 1000, 1023, 1024, 1025] * 4, 4),
    ("1kb", [1000] * 300, 4),
    ("mixed", list(rng.integers(1, 3000, 200)), 4),
    ("10kb", [10000, 9999, 16384, 16385, 20000], 4),
    ("alphabet8", list(rng.integers(1, 600, 100)), 7),
    ("alphabet16", list(rng.integers(1, 600, 100)), 13),
    ("bytes", list(rng.integers(1, 600, 100)), 200),
]:
    data = make(lengths, 0.01, nsym)
    t0 = time.time()
    b = N.Batch(ctx, *data, n_symbols=nsym)
    d = b.distances()
    t1 = time.time()
    ref = O.edit_distance_batch(*data)
    bad = np.nonzero(d != ref)[0]
    print(f"{name:16s} pairs={len(lengths):5d} gpu={t1-t0:.3f}s mismatches={len(bad)}", flush=True)
    if len(bad):
        ok = False
        for i in bad[:5]:
            print("   pair", i, "m,n=", data[2][i], data[4][i], "gpu", d[i], "oracle", ref[i])
    b.close()
# throughput: C2
n = 100000
a = rng.integers(0, 4, (n, 1000)).astype(np.uint8)
seqs = []
t0 = time.time()
dels = rng.random((n, 1000)) < 0.01
ins = rng.random((n, 1000)) < 0.01
insb = rng.integers(0, 4, (n, 1000)).astype(np.uint8)
parts = []
a_off = np.arange(n, dtype=np.uint64) * 1000
b_list = []
for i in range(n):
    row = a[i]
    keep = ~dels[i]
    # interleave: for each position optionally an inserted base before it
    tmp = np.empty(2000, np.uint8); cnt = 0
    idx_ins = ins[i]
    seq = np.stack([insb[i], row], 1).reshape(-1)
    mask = np.stack([idx_ins, keep], 1).reshape(-1)
    b_list.append(seq[mask])
b_len = np.array([len(x) for x in b_list], np.uint32)
b_off = (n * 1000 + np.concatenate([[0], np.cumsum(b_len)[:-1]])).astype(np.uint64)
syms = np.concatenate([a.reshape(-1)] + b_list)
print("gen", time.time() - t0)
bt = N.Batch(ctx, syms, a_off, np.full(n, 1000, np.uint32), b_off, b_len, 4)
d = bt.distances()
ref = O.edit_distance_batch(syms[:], a_off[:2000], np.full(2000, 1000, np.uint32), b_off[:2000], b_len[:2000])
print("C2 sample mismatches", int((d[:2000] != ref).sum()), "mean d", d.mean())
for it in (1, 10, 10):
    ms = bt.time_distances(it) / it
    cells = float((1000.0 * b_len).sum())
    print(f"C2: {ms:.3f} ms/launch  {n/ms/1e3:.2f} Mpairs/s  {cells/ms/1e9:.1f} TCUPS")
print("OK" if ok else "FAILED")
