"""Parsers, Variant serialisations and generators against vectors captured from the reference
(tests/golden/make_golden_variants.py): SURVEY 8(f) f3 / f4.  CPU only."""
import random

import pytest

from conftest import load_golden

GOLD = load_golden("variants.json")


def V(rec):
    from algebra_amd.variants import Variant
    return Variant(*rec)


def as_lists(variants):
    return [[v.start, v.end, v.sequence] for v in variants]


def check_outcome(rec, fn, *args):
    if "error" in rec:
        with pytest.raises(Exception) as exc:
            fn(*args)
        assert [type(exc.value).__name__, str(exc.value)] == rec["error"], rec
    else:
        assert as_lists(fn(*args)) == rec["variants"], rec


@pytest.mark.parametrize("rec", GOLD["parse_hgvs"], ids=lambda r: f"{r['expression']}|{'ref' if r['reference'] else ''}")
def test_parse_hgvs(rec):
    from algebra_amd.variants import parse_hgvs
    check_outcome(rec, parse_hgvs, rec["expression"], rec["reference"])


@pytest.mark.parametrize("rec", GOLD["parse_spdi"], ids=lambda r: str(r["expression"]))
def test_parse_spdi(rec):
    from algebra_amd.variants import parse_spdi
    check_outcome(rec, parse_spdi, rec["expression"])


def test_variant_serialisations():
    from algebra_amd.variants import patch, reverse_complement, to_hgvs
    for sequence, expected in GOLD["reverse_complement"]:
        assert reverse_complement(sequence) == expected
    for rec in GOLD["variant_to_hgvs"]:
        assert V(rec["variant"]).to_hgvs(rec["reference"], only_substitutions=rec["only_substitutions"]) == rec["hgvs"]
    for rec in GOLD["variant_to_spdi"]:
        assert V(rec["variant"]).to_spdi(rec["reference_id"]) == rec["spdi"]
    for rec in GOLD["variant_atomics"]:
        assert [as_lists(combo) for combo in V(rec["variant"]).atomics()] == rec["atomics"]
    for rec in GOLD["variant_reverse_complement"]:
        assert as_lists([V(rec["variant"]).reverse_complement(rec["pivot"])]) == [rec["result"]]
    for rec in GOLD["patch"]:
        assert patch(rec["reference"], [V(v) for v in rec["variants"]]) == rec["observed"]
    for rec in GOLD["to_hgvs"]:
        assert to_hgvs([V(v) for v in rec["variants"]], rec["reference"], sequence_prefix=rec.get("sequence_prefix", False)) == rec["hgvs"]


def test_misc_helpers():
    from algebra_amd.utils import fasta_sequence, slice_sequence, vcf_variant
    lines, sequence = GOLD["misc"]["fasta"]
    assert fasta_sequence(lines) == sequence
    line, variant = GOLD["misc"]["vcf"]
    assert as_lists([vcf_variant(line)]) == [variant]
    for sequence, positions, inversion, expected in GOLD["misc"]["slice"]:
        assert slice_sequence(sequence, positions, inversion) == expected
        assert slice_sequence(sequence, iter(positions), inversion) == expected


def test_generators_consume_the_random_stream_like_the_reference():
    """BASELINE config 1 draws its inputs with these generators: the same seed must give the same alleles."""
    from algebra_amd.utils import random_sequence, random_variants
    from algebra_amd.variants import parse_hgvs, to_hgvs
    case = GOLD["config1"]
    random.seed(case["seed"])
    reference = random_sequence(1000, 1000)
    assert reference == case["reference"]
    for rec in case["pairs"]:
        lhs = list(random_variants(reference, p=0.002))
        rhs = list(random_variants(reference, p=0.002))
        assert as_lists(lhs) == rec["lhs"] and as_lists(rhs) == rec["rhs"]
        assert to_hgvs(lhs, reference) == rec["lhs_hgvs"] and to_hgvs(rhs, reference) == rec["rhs_hgvs"]
        assert parse_hgvs(rec["lhs_hgvs"], reference) is not None               # the round trip is checked with the relations (GPU)
    random.seed(case["hot_seed"])
    from algebra_amd.variants import Variant
    for rec in case["hot_pairs"]:
        centre = random.choice([100, 400, 401, 402, 700])
        drawn = []
        for _ in range(2):
            window = reference[centre - 8:centre + 8]
            drawn.append([Variant(v.start + centre - 8, v.end + centre - 8, v.sequence) for v in random_variants(window, p=0.12)])
        assert as_lists(drawn[0]) == rec["lhs"] and as_lists(drawn[1]) == rec["rhs"]
        assert to_hgvs(drawn[0], reference) == rec["lhs_hgvs"] and to_hgvs(drawn[1], reference) == rec["rhs_hgvs"]
