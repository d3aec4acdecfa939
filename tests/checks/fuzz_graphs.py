"""Seeded fuzz of the graph paths against the oracle (dev tool): python tests/checks/fuzz_graphs.py [pairs] [seed]."""
import os, sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import gen
from algebra_amd import _native as N
from algebra_amd.lcs.encoding import encode_pairs
from oracle import oracle as O
n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rs = np.random.RandomState(seed)
pairs, shifts = [], []
kinds = ["random", "indel", "sparse", "tandem", "equal"]
for _ in range(n_pairs):
    kind = kinds[rs.randint(0, len(kinds))]
    length = int(rs.choice([rs.randint(0, 40), rs.randint(0, 200), rs.randint(0, 600)]))
    alphabet = "ACGT"[:rs.randint(1, 5)]
    if len(alphabet) == 1: length //= 6             # a homopolymer pair of 600 has half a million edges
    a, b = gen.pair_of_kind(rs, kind, length, alphabet)
    r = rs.randint(0, 8)
    if r == 0: b = b[: len(b) // 2]
    elif r == 1: a = a[len(a) // 3:]
    elif r == 2: b = b + a
    pairs.append((a, b)); shifts.append(int(rs.randint(0, 9)))
data = encode_pairs(pairs)
ctx = N.default_context()
results = []
for label, env in (("default", {}), ("sweep-all", {"MALCS_SWEEP_MIN_ROWS": "1", "MALCS_SWEEP_RATIO": "0"}), ("bitvector", {"MALCS_SWEEP_DMAX": "-1"})):
    for k in ("MALCS_SWEEP_MIN_ROWS", "MALCS_SWEEP_RATIO", "MALCS_SWEEP_DMAX"): os.environ.pop(k, None)
    os.environ.update(env)
    bt = N.Batch(ctx, *data)
    t0 = time.time(); res = bt.graphs(np.array(shifts, np.int64)); dt = time.time() - t0
    print(f"{label}: {n_pairs} pairs in {dt:.2f}s, {int(res.counts[:, 3].sum())} edges, {res.device_built} graphs built on the device", flush=True)
    results.append((label, res)); bt.close()
bad = 0
t0 = time.time()
for i, ((a, b), sh) in enumerate(zip(pairs, shifts)):          # one oracle graph alive at a time
    ref = O.lcs_graph(a, b, sh)
    for label, res in results:
        ok = (res.distance[i] == ref.distance and tuple(res.supremal[i]) == ref.supremal and
              np.array_equal(res.nodes(i), ref.nodes) and np.array_equal(res.edges(i), ref.edges))
        if not ok:
            bad += 1
            if bad <= 5: print("MISMATCH", label, i, repr(a[:80]), repr(b[:80]), len(a), len(b), flush=True)
print(f"checked against the oracle in {time.time()-t0:.1f}s: {bad} mismatches", flush=True)
sys.exit(1 if bad else 0)
