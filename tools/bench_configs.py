"""Timings of BASELINE configs 2-graph, 3 and 5 on one GPU (dev tool; numbers quoted in DESIGN.md)."""
import os, sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import gen
from algebra_amd import _native as N
from algebra_amd.lcs.encoding import encode_pairs
from algebra_amd.extractor import extract_sequence, to_hgvs

ctx = N.default_context()

def run_pairs(name, pairs, reps=2):
    syms, ao, al, bo, bl, ns = encode_pairs(pairs)
    bt = N.Batch(ctx, syms, ao, al, bo, bl, ns)
    cells = float((al.astype(np.float64) * bl).sum())
    for _ in range(reps):
        t0 = time.time(); g = bt.graphs(); wall = time.time() - t0
        tm = g.timing
        print(f"{name}: pairs={len(pairs)} cells={cells:.3g} wall={wall*1e3:.1f}ms fwd={tm[0]:.2f}ms reach={tm[1]:.2f}ms host={tm[2]:.1f}ms "
              f"d={g.distance.mean():.0f} nodes={g.counts[:,2].mean():.0f} edges={g.counts[:,3].mean():.0f}", flush=True)
        g.close()
    bt.close()

which = sys.argv[1:] or ["c2", "c3", "c5"]
if "c1" in which:
    # BASELINE config 1 (plumbing): random.Random(1)-style stream, 1 000 bp reference, 10 000 pairs of alleles from
    # random_variants(p=0.002), serialised to HGVS, re-parsed, compared
    import random
    from algebra_amd.relations import compare, compare_batch
    from algebra_amd.utils import random_sequence, random_variants
    from algebra_amd.variants import parse_hgvs, to_hgvs as allele_hgvs
    random.seed(1)
    reference = random_sequence(1000, 1000)
    text = [(allele_hgvs(list(random_variants(reference, p=0.002)), reference), allele_hgvs(list(random_variants(reference, p=0.002)), reference))
            for _ in range(10000)]
    t0 = time.time()
    lhs = [parse_hgvs(a, reference) for a, _ in text]; rhs = [parse_hgvs(b, reference) for _, b in text]
    t1 = time.time()
    rel = compare_batch(reference, lhs, rhs)
    t2 = time.time()
    one = [compare(reference, l, r) for l, r in zip(lhs[:300], rhs[:300])]
    t3 = time.time()
    assert one == rel[:300]
    from collections import Counter
    print(f"C1 10000 allele pairs: parse {1e3*(t1-t0):.0f} ms, compare_batch {1e3*(t2-t1):.0f} ms -> {10000/(t2-t0):.0f} relations/s end to end; "
          f"one compare() at a time: {1e3*(t3-t2)/300:.2f} ms/pair; {dict(Counter(r.value for r in rel))}", flush=True)
if "c2" in which:
    n = 100000
    data = gen.c2_batch(n)
    bt = N.Batch(ctx, *data, 4)
    for _ in range(2):
        t0 = time.time(); g = bt.graphs(); wall = time.time() - t0
        print(f"C2 graphs: pairs={n} wall={wall*1e3:.0f}ms fwd={g.timing[0]:.1f}ms reach={g.timing[1]:.1f}ms host={g.timing[2]:.0f}ms -> {n/wall:.0f} graphs/s", flush=True)
        g.close()
    bt.close()
if "c3" in which:
    pairs = []
    for seed in range(16):
        pairs.append(gen.tandem_pair(np.random.RandomState(300 + seed), 10000, 20))
    run_pairs("C3 16x10kb tandem", pairs)
    t0 = time.time()
    for a, b in pairs[:4]:
        v, g = extract_sequence(a, b); h = to_hgvs(v, a)
    print(f"C3 extract_sequence+to_hgvs (python facade, per pair): {(time.time()-t0)/4*1e3:.0f} ms", flush=True)
if "c5" in which:
    sparse = [gen.large_pair({"kind": "few", "seed": 50 + i, "length": 1000000, "events": 20}) for i in range(4)]
    run_pairs("C5(i) 4x1Mb sparse", sparse)
    t0 = time.time()
    for a, b in sparse:
        v, g = extract_sequence(a, b); h = to_hgvs(v, a)
    print(f"C5(i) extract_sequence+to_hgvs (python facade, str in, HGVS out, per pair): {(time.time()-t0)/4*1e3:.1f} ms", flush=True)
    rs = np.random.RandomState(5)
    a = gen.large_pair({"kind": "mutate", "seed": 5, "length": 1000000, "p": 0.0})[0]
    t0 = time.time(); b = gen.mutate(rs, a, 0.01, 0.01); print("gen", time.time() - t0)
    run_pairs("C5(ii) 1x1Mb 2%", [(a, b)], reps=1)
