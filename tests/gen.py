"""Seeded input generators shared by tests/golden/make_golden.py, the tests and bench.py.

Everything is driven by numpy's legacy RandomState (a frozen stream), so the same seed gives
the same sequences in the build container and on the GPU box.
"""

import numpy as np

DNA = "ACGT"


def random_seq(rs, length, alphabet=DNA):
    return "".join(alphabet[i] for i in rs.randint(0, len(alphabet), length))


def mutate(rs, seq, p_del, p_ins, alphabet=DNA):
    """SURVEY 8(d) C2 recipe: each position deleted w.p. p_del or preceded by a random symbol w.p. p_ins."""
    n = len(seq)
    dele = rs.random_sample(n) < p_del
    ins = rs.random_sample(n) < p_ins
    insb = rs.randint(0, len(alphabet), n)
    out = []
    for i, ch in enumerate(seq):
        if ins[i]:
            out.append(alphabet[insb[i]])
        if not dele[i]:
            out.append(ch)
    return "".join(out)


def tandem_pair(rs, length, n_tracts, indel=0.005, alphabet=DNA):
    """SURVEY 8(d) C3 recipe: tandem-repeat tracts whose copy number differs between the two."""
    ref, obs = [], []
    per = max(1, length // max(1, n_tracts))
    for _ in range(n_tracts):
        unit = random_seq(rs, rs.randint(1, 7), alphabet)
        copies = rs.randint(5, 31)
        delta = rs.randint(1, 6) * (1 if rs.randint(0, 2) else -1)
        flank = random_seq(rs, max(0, per - len(unit) * copies), alphabet)
        ref.append(flank + unit * copies)
        obs.append(mutate(rs, flank, indel, indel, alphabet) + unit * max(0, copies + delta))
    return "".join(ref), "".join(obs)


def pair_of_kind(rs, kind, length, alphabet):
    a = random_seq(rs, length, alphabet)
    if kind == "random":
        b = random_seq(rs, max(0, length + rs.randint(-3, 4)), alphabet)
    elif kind == "indel":
        b = mutate(rs, a, 0.1, 0.1, alphabet)
    elif kind == "sparse":
        b = mutate(rs, a, 0.02, 0.02, alphabet)
    elif kind == "tandem":
        unit = random_seq(rs, rs.randint(1, 4), alphabet)
        k = rs.randint(2, 6)
        pos = rs.randint(0, length + 1)
        a = a[:pos] + unit * k + a[pos:]
        b = a[:pos] + unit * max(0, k + rs.randint(-2, 3)) + a[pos + len(unit) * k:]
    elif kind == "equal":
        b = a
    else:
        raise ValueError(kind)
    return a, b


def small_cases():
    rs = np.random.RandomState(1001)
    cases = []
    alphabets = ["A", "AC", "ACG", "ACGT"]
    kinds = ["random", "indel", "sparse", "tandem", "equal"]
    for i in range(600):
        alphabet = alphabets[rs.randint(0, 4)]
        kind = kinds[rs.randint(0, len(kinds))]
        length = int(rs.randint(0, 41))
        a, b = pair_of_kind(rs, kind, length, alphabet)
        if i % 17 == 0:
            a = ""
        if i % 23 == 0:
            b = ""
        cases.append({"reference": a, "observed": b, "shift": [0, 0, 3, 5][rs.randint(0, 4)]})
    return cases


def medium_cases():
    rs = np.random.RandomState(2002)
    cases = []
    for length in (63, 64, 65, 127, 128, 129, 255, 256, 257, 511, 512, 513, 1000, 1023, 1024, 1025):
        for name, p in (("0.5%", 0.0025), ("2%", 0.01), ("10%", 0.05)):
            a = random_seq(rs, length)
            cases.append({"name": f"len{length}-{name}", "reference": a, "observed": mutate(rs, a, p, p), "shift": 0})
    for length in (64, 128, 1024):
        a = random_seq(rs, length)
        cases.append({"name": f"len{length}-equal", "reference": a, "observed": a, "shift": 7})
        cases.append({"name": f"len{length}-random", "reference": a, "observed": random_seq(rs, length), "shift": 0})
        cases.append({"name": f"len{length}-vs-empty", "reference": a, "observed": "", "shift": 0})
        cases.append({"name": f"len{length}-homopolymer", "reference": "A" * length, "observed": "A" * (length - 3), "shift": 2})
    for length, alphabet in ((300, "AC"), (300, "ACGTN"), (500, "ACGTNRYKM"), (200, "ABCDEFGHIJKLMNOPQRSTUVWXYZ")):
        a = random_seq(rs, length, alphabet)
        cases.append({"name": f"len{length}-sigma{len(alphabet)}", "reference": a,
                      "observed": mutate(rs, a, 0.03, 0.03, alphabet), "shift": 0})
    for length, tracts in ((600, 3), (1500, 6), (2048, 8), (4096, 10)):
        a, b = tandem_pair(rs, length, tracts)
        cases.append({"name": f"tandem{length}", "reference": a, "observed": b, "shift": 11})
    for length in (2048, 4096):
        a = random_seq(rs, length)
        cases.append({"name": f"len{length}-1%", "reference": a, "observed": mutate(rs, a, 0.005, 0.005), "shift": 0})
    return cases


def max_distance_cases():
    rs = np.random.RandomState(3003)
    cases = []
    for length in (10, 30, 100):
        a = random_seq(rs, length)
        b = mutate(rs, a, 0.1, 0.1)
        cases.append({"reference": a, "observed": b, "limits": [0, 1, 2, 3, 5, 8, 13, 21, 34, 1000]})
    return cases


def sequence_cases():
    rs = np.random.RandomState(4004)
    cases = []
    for i in range(120):
        length = int(rs.randint(1, 200))
        kind = ["indel", "sparse", "tandem", "random"][rs.randint(0, 4)]
        if kind == "random":
            length = min(length, 40)
        a, b = pair_of_kind(rs, kind, length, DNA)
        cases.append({"name": f"{kind}{length}-{i}", "reference": a, "observed": b})
    for length, tracts in ((400, 3), (800, 5), (1200, 6)):
        a, b = tandem_pair(rs, length, tracts)
        cases.append({"name": f"tandem{length}", "reference": a, "observed": b})
    return cases


def relation_cases():
    rs = np.random.RandomState(5005)
    cases = []
    for _ in range(150):
        length = int(rs.randint(5, 80))
        r = random_seq(rs, length)
        pick = rs.randint(0, 4)
        if pick == 0:
            a = mutate(rs, r, 0.05, 0.05)
            b = mutate(rs, r, 0.05, 0.05)
        elif pick == 1:
            a = mutate(rs, r, 0.05, 0.05)
            b = mutate(rs, a, 0.03, 0.03)
        elif pick == 2:
            a = mutate(rs, r, 0.05, 0.05)
            b = a
        else:
            a = mutate(rs, r, 0.1, 0.0)
            b = mutate(rs, r, 0.0, 0.1)
        cases.append({"reference": r, "lhs": a, "rhs": b})
    return cases


def large_cases():
    return [
        {"name": "c3-10kb-tandem-a", "kind": "tandem", "seed": 3, "length": 10000, "tracts": 20},
        {"name": "c3-10kb-tandem-b", "kind": "tandem", "seed": 33, "length": 10000, "tracts": 20},
        {"name": "10kb-2%", "kind": "mutate", "seed": 7, "length": 10000, "p": 0.005},
        {"name": "100kb-sparse", "kind": "few", "seed": 8, "length": 100000, "events": 20},
        {"name": "c5-1mb-sparse", "kind": "few", "seed": 5, "length": 1000000, "events": 20},
    ]


def large_pair(case):
    rs = np.random.RandomState(case["seed"])
    if case["kind"] == "tandem":
        return tandem_pair(rs, case["length"], case["tracts"])
    codes = rs.randint(0, 4, case["length"]).astype(np.uint8)
    a = np.frombuffer(b"ACGT", dtype=np.uint8)[codes].tobytes().decode()
    if case["kind"] == "mutate":
        return a, mutate(rs, a, case["p"], case["p"])
    # "few": a handful of 1-3 bp indels
    pos = sorted(int(x) for x in rs.randint(0, case["length"], case["events"]))
    out, last = [], 0
    for p in pos:
        if p < last:
            continue
        out.append(a[last:p])
        k = int(rs.randint(1, 4))
        if rs.randint(0, 2):
            last = min(len(a), p + k)
        else:
            out.append(random_seq(rs, k))
            last = p
    out.append(a[last:])
    return a, "".join(out)


def c2_batch(n_pairs, seed=2, length=1000, p=0.01):
    """BASELINE config 2 as flat arrays: (syms, a_off, a_len, b_off, b_len), codes 0..3."""
    rs = np.random.RandomState(seed)
    a = rs.randint(0, 4, (n_pairs, length)).astype(np.uint8)
    dele = rs.random_sample((n_pairs, length)) < p
    ins = rs.random_sample((n_pairs, length)) < p
    insb = rs.randint(0, 4, (n_pairs, length)).astype(np.uint8)
    seq = np.stack([insb, a], 2).reshape(n_pairs, 2 * length)
    mask = np.stack([ins, ~dele], 2).reshape(n_pairs, 2 * length)
    b_len = mask.sum(1).astype(np.uint32)
    b_flat = seq[mask]
    a_off = (np.arange(n_pairs, dtype=np.uint64) * np.uint64(length))
    b_off = (np.uint64(n_pairs * length) + np.concatenate([[0], np.cumsum(b_len[:-1], dtype=np.uint64)])).astype(np.uint64)
    syms = np.concatenate([a.reshape(-1), b_flat])
    return syms, a_off, np.full(n_pairs, length, np.uint32), b_off, b_len


def window_mix(n_pairs, seed=7, nsym=4):
    """Pairs for the window class of family 1 with mixed lengths (200 ... 4 000), indel rates (0 ... 4 %) and
    length differences (0 ... 60 on either end), either side first: tiles then mix lanes whose masked first /
    last blocks and plain stretches differ.  Flat arrays like c2_batch."""
    rng = np.random.default_rng(seed)
    seqs = []
    for _ in range(n_pairs):
        length = int(rng.choice([rng.integers(200, 420), rng.integers(900, 1100), rng.integers(200, 4000)]))
        a = rng.integers(0, nsym, length).astype(np.uint8)
        p = float(rng.choice([0.0, 0.002, 0.01, 0.02, 0.04]))
        keep = rng.random(length) >= p
        ins = rng.random(length) < p
        b = np.stack([np.where(ins, rng.integers(0, nsym, length), 255).astype(np.uint8),
                      np.where(keep, a, 255).astype(np.uint8)], 1).reshape(-1)
        b = b[b != 255]
        pad = rng.integers(0, nsym, int(rng.choice([0, 0, 1, 5, 31, 32, 33, 60]))).astype(np.uint8)
        side = rng.integers(0, 3)
        b = np.concatenate([pad, b]) if side == 0 else (np.concatenate([b, pad]) if side == 1 else b)
        seqs.append((a, b) if rng.random() < 0.5 else (b, a))
    syms = np.concatenate([x for ab in seqs for x in ab] + [np.zeros(1, np.uint8)])
    lens = np.array([[len(a), len(b)] for a, b in seqs], dtype=np.uint32)
    offs = np.concatenate([[0], np.cumsum(lens.reshape(-1).astype(np.uint64))[:-1]]).astype(np.uint64).reshape(-1, 2)
    return (syms, np.ascontiguousarray(offs[:, 0]), np.ascontiguousarray(lens[:, 0]),
            np.ascontiguousarray(offs[:, 1]), np.ascontiguousarray(lens[:, 1]))


def clustered_variants(rs, reference, n_clusters, per_cluster, width):
    """SURVEY 8(d) C4 recipe in small: single-variant alleles packed into hot spots (uniform placement
    would make every pair trivially disjoint).  Returns a list of (start, end, inserted)."""
    out = []
    for _ in range(n_clusters):
        centre = int(rs.randint(width, len(reference) - width))
        for _ in range(per_cluster):
            pos = centre + int(rs.randint(-width // 2, width // 2))
            kind = rs.randint(0, 3)
            if kind == 0:                       # deletion
                out.append((pos, pos + int(rs.randint(1, 4)), ""))
            elif kind == 1:                     # insertion
                out.append((pos, pos, random_seq(rs, int(rs.randint(1, 4)))))
            else:                               # substitution / delins
                k = int(rs.randint(1, 3))
                out.append((pos, pos + k, random_seq(rs, int(rs.randint(1, 3)))))
    return out


def matrix_case(seed=6006, length=400, n_clusters=5, per_cluster=10, width=16, tandem=True):
    rs = np.random.RandomState(seed)
    reference = random_seq(rs, length)
    if tandem:                                   # half of the hot spots sit on tandem repeats
        pieces, pos = [], 0
        for k in range(n_clusters // 2 + 1):
            cut = (k + 1) * length // (n_clusters // 2 + 2)
            unit = random_seq(rs, int(rs.randint(1, 4)))
            pieces.append(reference[pos:cut] + unit * int(rs.randint(4, 9)))
            pos = cut
        pieces.append(reference[pos:])
        reference = "".join(pieces)
    return {"reference": reference, "variants": clustered_variants(rs, reference, n_clusters, per_cluster, width)}
