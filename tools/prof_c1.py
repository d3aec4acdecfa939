"""cProfile of BASELINE config 1 (10 000 allele pairs from HGVS text to relations) (dev tool)."""
import sys, time, random, cProfile, pstats
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from algebra_amd.relations import compare_batch
from algebra_amd.utils import random_sequence, random_variants
from algebra_amd.variants import parse_hgvs, to_hgvs as allele_hgvs
random.seed(1)
reference = random_sequence(1000, 1000)
text = [(allele_hgvs(list(random_variants(reference, p=0.002)), reference), allele_hgvs(list(random_variants(reference, p=0.002)), reference))
        for _ in range(10000)]
lhs = [parse_hgvs(a, reference) for a, _ in text]; rhs = [parse_hgvs(b, reference) for _, b in text]
compare_batch(reference, lhs[:100], rhs[:100])
pr = cProfile.Profile(); pr.enable()
t0 = time.time(); rel = compare_batch(reference, lhs, rhs); t1 = time.time()
pr.disable()
print("compare_batch", t1 - t0)
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
