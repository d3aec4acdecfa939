"""Quick GPU check of family 2 against the oracle (dev tool)."""
import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
from algebra_amd import _native as N
from oracle import oracle as O
import gen

def pack(pairs):
    enc = []
    for a, b in pairs:
        enc.extend(O.encode(a, b))
    syms = np.concatenate(enc + [np.zeros(1, np.uint8)])
    off = np.concatenate([[0], np.cumsum([len(x) for x in enc])]).astype(np.uint64)
    lens = np.array([len(x) for x in enc], np.uint32)
    return syms, off[0:-1:2].copy(), lens[0::2].copy(), off[1:-1:2].copy() if len(enc) > 1 else off[1:2], lens[1::2].copy()

ctx = N.Context(0)
rs = np.random.RandomState(7)
groups = {
  "small": [(c["reference"], c["observed"], c["shift"]) for c in gen.small_cases()],
  "medium": [(c["reference"], c["observed"], c["shift"]) for c in gen.medium_cases() if set(c["reference"] + c["observed"]) <= set("ACGT")],
}
ok = True
for name, cases in groups.items():
    pairs = [(a, b) for a, b, _ in cases]
    shifts = np.array([s for _, _, s in cases], np.int64)
    syms, ao, al, bo, bl = pack(pairs)
    bt = N.Batch(ctx, syms, ao, al, bo, bl, 4)
    t0 = time.time()
    g = bt.graphs(shift=shifts)
    t1 = time.time()
    bad = 0
    for i, (a, b, s) in enumerate(cases):
        ref = O.lcs_graph(a, b, s)
        same = (g.distance[i] == ref.distance and tuple(g.supremal[i]) == ref.supremal
                and np.array_equal(g.nodes(i), ref.nodes) and np.array_equal(g.edges(i), ref.edges))
        if not same:
            bad += 1
            if bad <= 3:
                print("MISMATCH", name, i, repr(a), repr(b), s)
                print("  gpu ", g.distance[i], g.supremal[i], g.nodes(i).tolist()[:8], g.runs(i).tolist()[:10])
                print("  orc ", ref.distance, ref.supremal, ref.nodes.tolist()[:8], ref.level_lists()[:10])
    print(f"{name}: {len(cases)} cases, {bad} mismatches, {t1-t0:.3f}s timing(ms) {g.timing}", flush=True)
    ok &= bad == 0
print("OK" if ok else "FAILED")
