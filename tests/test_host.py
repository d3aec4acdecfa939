"""CPU-side tests of the product: the C-ABI library loads and exports what include/malcs.h
declares, fails loudly without a GPU, and the host-side mirrors (Variant, patch, trim, encoding)
behave like the reference (vectors from tests/golden)."""

import os
import re

import numpy as np
import pytest

from conftest import ROOT, load_golden

REF = load_golden("reference_tests.json")


def header_functions():
    text = open(os.path.join(ROOT, "include", "malcs.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(malcs_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    import ctypes
    from algebra_amd import _native
    lib = ctypes.CDLL(_native.LIB_PATH)
    names = header_functions()
    assert len(names) >= 20
    for name in names:
        assert hasattr(lib, name), name
    assert set(_native.EXPORTS) == set(names)
    assert _native.lib.malcs_version() == 100


def test_no_cpu_fallback_without_device():
    import ctypes
    from algebra_amd import _native
    n = ctypes.c_int(-1)
    rc = _native.lib.malcs_device_count(ctypes.byref(n))
    if rc == 0 and n.value > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(_native.MalcsError):
        _native.Context(0)
    import algebra_amd
    with pytest.raises(_native.MalcsError):
        algebra_amd.edit_distance("ACGT", "AGT")


def test_product_does_not_import_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "algebra_amd")):
        for name in files:
            if name.endswith((".py", ".cpp", ".hip", ".hpp", ".h")) or name == "Makefile":
                text = open(os.path.join(dirpath, name)).read()
                assert "oracle" not in text.lower(), (dirpath, name)


def test_trim_vectors():
    from algebra_amd.lcs import trim
    for rec in REF["trim"]:
        assert list(trim(rec["lhs"], rec["rhs"])) == rec["trim"]
    for rec in REF["sequences"]:
        assert list(trim(rec["reference"], rec["observed"])) == rec["trim"]
    a = "ACGT" * 50000
    b = a[:70001] + "T" + a[70001:]
    assert trim(a, b) == (70001, len(a) - 70001)
    assert trim(a, a) == (len(a), 0)


def test_variant_value_type():
    from algebra_amd.variants import Variant, patch, to_hgvs
    v = Variant(3, 5, "AC")
    assert v == Variant(3, 5, "AC") and hash(v) == hash(Variant(3, 5, "AC"))
    assert len(v) == 4 and repr(v) == "3:5/AC"
    assert Variant(0, 1, "") < Variant(2, 3, "")
    with pytest.raises(ValueError, match="unorderable variants"):
        Variant(0, 2, "") < Variant(1, 3, "")
    with pytest.raises(ValueError, match="unorderable variants"):
        sorted([Variant(1, 1, "A"), Variant(1, 1, "C")])
    assert patch("ACGT", [Variant(1, 2, "TT"), Variant(3, 3, "G")]) == "ATTGGT"
    assert patch("ACGT", []) == "ACGT"
    assert Variant(1, 2, "").is_disjoint(Variant(2, 3, ""))
    assert not Variant(1, 3, "").is_disjoint(Variant(2, 4, ""))
    assert not Variant(2, 2, "A").is_disjoint(Variant(2, 2, "AC"))
    assert Variant(2, 2, "A").is_disjoint(Variant(2, 2, "C"))
    assert Variant(2, 2, "A").is_disjoint(Variant(4, 4, "A"))
    assert to_hgvs([], "ACGT") == "=" and Variant(1, 2, "T").to_hgvs("ACGT") == "2C>T"
    assert Variant(1, 3, "").to_hgvs() == "2_3del" and Variant(2, 2, "GG").to_hgvs() == "2_3insGG"
    with pytest.raises(TypeError):
        Variant.create_safe("1", 2, "")
    with pytest.raises(ValueError):
        Variant.create_safe(3, 2, "")


def test_variant_atomics_order():
    from algebra_amd.variants import Variant
    got = [[(a.start, a.end, a.sequence) for a in alt] for alt in Variant(0, 2, "AA").atomics()]
    # order pinned by the reference's test_bfs_traversal_atomics vector for Variant(0, 2, "AA")
    assert got[0] == [(0, 0, "AA"), (0, 1, ""), (1, 2, "")]
    assert got[1] == [(0, 0, "A"), (0, 1, ""), (1, 1, "A"), (1, 2, "")]
    assert got[2] == [(0, 0, "A"), (0, 1, ""), (1, 2, ""), (2, 2, "A")]
    assert got[3] == [(0, 1, ""), (1, 1, "AA"), (1, 2, "")]
    assert got[5] == [(0, 1, ""), (1, 2, ""), (2, 2, "AA")]
    assert len(got) == 6
    assert list(Variant(3, 3, "").atomics()) == [[]]


def test_encoding():
    from algebra_amd.lcs.encoding import encode_pairs
    syms, a_off, a_len, b_off, b_len, n = encode_pairs([("ACGT", "TGCA"), ("ACGT", "")])
    assert n == 4 and syms[:8].tolist() == [0, 1, 2, 3, 3, 2, 1, 0]
    assert a_off.tolist() == [0, 0] and b_len.tolist() == [4, 0]
    syms, *_, n = encode_pairs([("AXB", "B中X")])
    assert n == 7 and len(set(syms[:6].tolist())) == 4 and syms[1] == syms[5] and syms[2] == syms[3]
    syms, *_, n = encode_pairs([("", "")])
    assert n == 4
    # more than 256 distinct symbols in a batch: only the symbols the two sides of a pair share need codes of their own
    # (equality is all the reference looks at, edit_distance.py:26); the rest get one code per side
    wide_a = "".join(chr(300 + i) for i in range(300))
    syms, a_off, a_len, b_off, b_len, n = encode_pairs([(wide_a, "A" + wide_a[5:9] + "B"), ("XY", "YZ")])
    assert n == 7                                             # 4 shared in pair 0, 1 in pair 1, + one code per side
    a0 = syms[int(a_off[0]):int(a_off[0]) + 300]
    b0 = syms[int(b_off[0]):int(b_off[0]) + 6]
    assert (a0[5:9] == b0[1:5]).all() and len(set(a0[5:9].tolist())) == 4
    assert set(np.delete(a0, [5, 6, 7, 8]).tolist()) == {5} and b0[0] == b0[5] == 6
    from algebra_amd.lcs.encoding import TooManySymbols
    with pytest.raises(TooManySymbols):
        encode_pairs([(wide_a, wide_a[::-1])])                # 300 shared symbols in one pair: nothing left to fold
