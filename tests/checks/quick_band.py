"""Quick GPU check of the sliding-window distance kernel on long pairs (dev tool)."""
import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import gen
import algebra_amd
from oracle import oracle as O
import json
LARGE = json.load(open("tests/golden/large.json"))["from_sequence"]
rs = np.random.RandomState(11)
pairs = []
for rec in LARGE:
    if rec["len_reference"] > 16000:
        a, b = gen.large_pair(rec["case"]); pairs.append((a, b, rec["distance"]))
for m, p in ((70000, 0.001), (100000, 0.01), (200000, 0.003), (300000, 0.02), (66000, 0.08)):
    codes = rs.randint(0, 4, m).astype(np.uint8)
    a = np.frombuffer(b"ACGT", dtype=np.uint8)[codes].tobytes().decode()
    b = gen.mutate(rs, a, p, p)
    pairs.append((a, b, None))
a = pairs[-1][0]
pairs.append((a, a[:-5000], None)); pairs.append((a[3000:], a, None)); pairs.append((a, a, None))
for a, b, want in pairs:
    t0 = time.time(); d = algebra_amd.edit_distance(a, b); t1 = time.time()
    if want is None:
        want = O.edit_distance(a, b)
    t2 = time.time()
    print(f"m={len(a)} n={len(b)} gpu={d} want={want} {'OK' if d == want else 'MISMATCH'} gpu {t1-t0:.3f}s oracle {t2-t1:.3f}s", flush=True)
