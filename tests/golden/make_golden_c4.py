"""Golden fixture for BASELINE config 4 at FULL size, from the real reference (build container only):

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python3 /root/repo/tests/golden/make_golden_c4.py

The 5 000 x 5 000 `graph_based.compare` matrix over clustered single-variant alleles in a 100 kb locus
(gen.matrix_case(seed=4, length=100000, n_clusters=50, per_cluster=100, width=200), the workload bench.py times).
The reference computes every pair i < j (12 497 500 calls, about a minute here); stored are the SHA-256 of that upper
triangle (row-major, one byte per label in the order of `Relation`), its histogram, the supremals' and distances'
digests, and 3 000 explicit labels below the diagonal (the reference called with the arguments swapped) that pin the
mirror rule.  Data only: inputs are regenerated from the seed by tests/gen.py.
"""

import hashlib
import json
import os
import sys
import time

REFERENCE = os.environ.get("ALGEBRA_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REFERENCE)
sys.path.insert(0, os.path.join(HERE, ".."))

import numpy as np  # noqa: E402
from algebra import Relation  # noqa: E402
from algebra.lcs import LCSgraph  # noqa: E402
from algebra.relations import graph_based  # noqa: E402
from algebra.variants import Variant  # noqa: E402

import gen  # noqa: E402  (tests/gen.py)

PARAMS = {"seed": 4, "length": 100000, "n_clusters": 50, "per_cluster": 100, "width": 200}


def digest(obj):
    return hashlib.sha256(json.dumps(obj, separators=(",", ":"), sort_keys=True).encode()).hexdigest()


def main():
    t0 = time.time()
    case = gen.matrix_case(**PARAMS)
    reference = case["reference"]
    graphs = [LCSgraph.from_variant(reference, [Variant(*v)]) for v in case["variants"]]
    n = len(graphs)
    print(f"{n} supremal graphs from the reference ({time.time() - t0:.1f}s)", flush=True)
    codes = {relation: code for code, relation in enumerate(Relation)}
    upper = np.zeros(n * (n - 1) // 2, dtype=np.uint8)
    at = 0
    for i in range(n):
        a = graphs[i]
        for j in range(i + 1, n):
            upper[at] = codes[graph_based.compare(reference, a, graphs[j])]
            at += 1
        if i % 500 == 0:
            print(f"  row {i} ({time.time() - t0:.1f}s)", flush=True)
    rs = np.random.RandomState(44)
    below = []
    # half of the sample inside the clusters (where the interesting labels are), half anywhere
    per = PARAMS["per_cluster"]
    while len(below) < 3000:
        if len(below) % 2 == 0:
            c = int(rs.randint(0, PARAMS["n_clusters"]))
            i, j = sorted(int(x) for x in rs.randint(c * per, (c + 1) * per, 2))
        else:
            i, j = sorted(int(x) for x in rs.randint(0, n, 2))
        if i == j:
            continue
        below.append([j, i, codes[graph_based.compare(reference, graphs[j], graphs[i])]])
    out = {"params": PARAMS, "n": n, "input_sha256": digest([reference, [list(v) for v in case["variants"]]]),
           "supremals_sha256": digest([[g.supremal.start, g.supremal.end, g.supremal.sequence] for g in graphs]),
           "distances_sha256": digest([g.distance for g in graphs]),
           "upper_sha256": hashlib.sha256(upper.tobytes()).hexdigest(),
           "upper_histogram": np.bincount(upper, minlength=5).tolist(),
           "below_diagonal_sample": below,
           "reference_seconds": round(time.time() - t0, 1)}
    with open(os.path.join(HERE, "c4_matrix.json"), "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print(f"c4_matrix.json: histogram {out['upper_histogram']} ({time.time() - t0:.1f}s)")


if __name__ == "__main__":
    main()
