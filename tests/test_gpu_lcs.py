"""GPU parity tests: the HIP path (through the C-ABI and the algebra.lcs-shaped facade) against the
golden fixtures captured from the reference and against the CPU oracle on seeded inputs."""

import os
import numpy as np
import pytest

from conftest import load_golden
from util import assert_graph, digest, graph_record, graph_record_from_arrays, observables
import gen

pytestmark = pytest.mark.gpu

REF = load_golden("reference_tests.json")
SMALL = load_golden("random_small.json")
MEDIUM = load_golden("random_medium.json")
SEQS = load_golden("random_sequences.json")
LARGE = load_golden("large.json")


@pytest.fixture(scope="module")
def native():
    from algebra_amd import _native
    assert _native.default_context().device_info()[1] > 0      # the HIP library is the one doing the work
    return _native


# ---------------------------------------------------------------------------------------------
# family 1: distances
# ---------------------------------------------------------------------------------------------

def test_edit_distance_reference_vectors(native):
    import algebra_amd
    # the reference's own three vectors, as its test reads (tests/lcs/test_edit_distance.py:5-11)
    assert algebra_amd.edit_distance("CTCGGCATTA", "GGCTGGCTGT") == 6
    assert algebra_amd.edit_distance("TTT", "TTTTAT") == 3
    assert algebra_amd.edit_distance("TTTTAT", "TTT") == 3
    cases = REF["lcs"] + SMALL["lcs"] + MEDIUM["lcs"]
    got = algebra_amd.edit_distance_batch([(c["reference"], c["observed"]) for c in cases])
    assert got.tolist() == [c["edit_distance"] for c in cases]
    swapped = algebra_amd.edit_distance_batch([(c["observed"], c["reference"]) for c in cases])
    assert swapped.tolist() == got.tolist()


@pytest.mark.parametrize("n_symbols", [1, 2, 4, 5, 8, 9, 16, 17, 200])
def test_edit_distance_vs_oracle_word_boundaries(native, n_symbols):
    from oracle import oracle as O
    rs = np.random.RandomState(100 + n_symbols)
    lengths = [0, 1, 31, 32, 33, 63, 64, 65, 127, 128, 129, 191, 192, 193, 255, 256, 257, 300, 511, 512, 513,
               1000, 1023, 1024, 1025, 2047, 2048, 2049, 4095, 4096, 4097]
    seqs = []
    for m in lengths:
        a = rs.randint(0, n_symbols, m).astype(np.uint8)
        keep = rs.random_sample(m) > 0.02
        b = np.concatenate([a[keep], rs.randint(0, n_symbols, rs.randint(0, 4)).astype(np.uint8)])
        seqs.append((a, b))
        seqs.append((a, rs.randint(0, n_symbols, max(0, m + rs.randint(-2, 3))).astype(np.uint8)))
    flat = [x for ab in seqs for x in ab]
    off = np.concatenate([[0], np.cumsum([len(x) for x in flat])]).astype(np.uint64)
    lens = np.array([len(x) for x in flat], np.uint32)
    syms = np.concatenate(flat + [np.zeros(1, np.uint8)])
    args = (syms, off[0:-1:2], lens[0::2], off[1::2], lens[1::2])
    batch = native.Batch(native.default_context(), *args, n_symbols=max(n_symbols, 1))
    got = batch.distances()
    batch.close()
    assert got.tolist() == O.edit_distance_batch(*args).tolist()


def test_edit_distance_long_patterns(native):
    from oracle import oracle as O
    rs = np.random.RandomState(5)
    pairs = []
    for m in (10000, 16383, 16384, 16385, 40000, 65536):
        a = gen.random_seq(rs, m)
        pairs.append((a, gen.mutate(rs, a, 0.003, 0.003)))
    import algebra_amd
    got = algebra_amd.edit_distance_batch(pairs)
    assert got.tolist() == [O.edit_distance(a, b) for a, b in pairs]


def use_path(monkeypatch, path):
    """Long pairs have two device paths with identical results: the output-sensitive sweeps (default for similar
    pairs) and the bit-vector band kernels (what remains for divergent ones); "band" switches the sweeps off."""
    if path == "band":
        monkeypatch.setenv("MALCS_SWEEP_DMAX", "-1")
        monkeypatch.setenv("MALCS_SWEEP_DCAP", "0")
    elif path == "sweep-all":                  # also the shortest windows through the one-wave sweeps
        monkeypatch.setenv("MALCS_SWEEP_MIN_ROWS", "1")
        monkeypatch.setenv("MALCS_SWEEP_RATIO", "0")       # ... and the divergent ones
    elif path == "mixed":                      # sweeps give up early: their results and the band kernels' in one batch
        monkeypatch.setenv("MALCS_SWEEP_DMAX", "700")
        monkeypatch.setenv("MALCS_SWEEP_DCAP", "700")


@pytest.mark.parametrize("path", ["sweep", "band", "mixed"])
def test_edit_distance_sliding_window(native, path, monkeypatch):
    """Sequences longer than a resident pattern (> 65 536): furthest-reaching sweep / sliding-window kernel,
    incl. the 100 kb and 1 Mb pairs whose distances were captured from the reference."""
    use_path(monkeypatch, path)
    from oracle import oracle as O
    import algebra_amd
    pairs, want = [], []
    for rec in LARGE["from_sequence"]:
        if rec["len_reference"] > 16000:
            a, b = gen.large_pair(rec["case"])
            assert digest([a, b]) == rec["input_sha256"]
            pairs.append((a, b))
            want.append(rec["distance"])
    rs = np.random.RandomState(11)
    for m, p in ((70000, 0.001), (100000, 0.01), (200000, 0.003), (300000, 0.02), (66000, 0.08)):
        a = gen.large_pair({"kind": "mutate", "seed": m, "length": m, "p": 0.0})[0]
        b = gen.mutate(rs, a, p, p)
        pairs.append((a, b))
        want.append(O.edit_distance(a, b))
    a = pairs[-1][0]
    for x, y in ((a, a[:-5000]), (a[3000:], a), (a, a), (a, a[:100] + a[40000:])):
        pairs.append((x, y))
        want.append(O.edit_distance(x, y))
    got = algebra_amd.edit_distance_batch(pairs)          # one batch: mixes window widths
    assert got.tolist() == want
    short = ("ACGT" * 20, "ACGA" * 20)
    mixed = algebra_amd.edit_distance_batch([pairs[0], short, pairs[3]])
    assert mixed.tolist() == [want[0], O.edit_distance(*short), want[3]]


def test_edit_distance_c2_full_size(native):
    """BASELINE config 2 at full size: 100k pairs of 1 kb windows, ~2% indels."""
    from oracle import oracle as O
    n = 100000
    syms, a_off, a_len, b_off, b_len = gen.c2_batch(n)
    ctx = native.default_context()
    batch = native.Batch(ctx, syms, a_off, a_len, b_off, b_len, 4)
    d = batch.distances()
    batch.close()
    # size-independent properties of the indel distance
    m, k = a_len.astype(np.int64), b_len.astype(np.int64)
    assert np.all(d >= np.abs(m - k)) and np.all(d <= m + k)
    assert np.all((d - (m + k)) % 2 == 0)
    # symmetry: the same batch with the roles swapped
    swapped = native.Batch(ctx, syms, b_off, b_len, a_off, a_len, 4)
    assert np.array_equal(swapped.distances(), d)
    swapped.close()
    # identical pairs give 0
    same = native.Batch(ctx, syms, a_off[:1000], a_len[:1000], a_off[:1000], a_len[:1000], 4)
    assert not same.distances().any()
    same.close()
    # a seeded sample against the oracle
    pick = np.random.RandomState(0).choice(n, 3000, replace=False)
    ref = O.edit_distance_batch(syms, a_off[pick], a_len[pick], b_off[pick], b_len[pick])
    assert np.array_equal(d[pick], ref)
    assert 15 < d.mean() < 25


def test_edit_distance_c2_full_size_every_kernel_agrees(native, monkeypatch):
    """BASELINE config 2 at full size through every kernel that can take it: the window class as the plan picks it (two
    limbs), forced to three limbs (LDS stream, and the plane-folding paths), and the full-matrix kernel alone -- four
    instruction streams, one answer, and that answer is the oracle's on all 100 000 pairs."""
    from oracle import oracle as O
    data = gen.c2_batch(100000)
    ref = O.edit_distance_batch(*data)
    for env in ({}, {"MALCS_DIAG_HALF": "0"}, {"MALCS_DIAG_HALF": "0", "MALCS_DIAG_LDS3": "0"}, {"MALCS_DIAG_NLW": "0"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        batch = native.Batch(native.default_context(), *data, n_symbols=4)
        info = batch.plan_info()
        got = batch.distances()
        batch.close()
        for k in env:
            monkeypatch.delenv(k)
        assert np.array_equal(got, ref), env
        if "MALCS_DIAG_NLW" in env:
            assert info["window_tiles"] == 0
        else:
            assert info["window_tiles"] == 1563 and (info["two_limb_tiles"] == 0) == ("MALCS_DIAG_HALF" in env)


def _window_class_batch(rs, n_pairs, nsym):
    """Pairs long enough for the window class, divergence from identical to unrelated, length gaps up to 300."""
    seqs = []
    for i in range(n_pairs):
        length = int(rs.randint(250, 2600))
        a = rs.randint(0, nsym, length).astype(np.uint8)
        kind = i % 8
        if kind == 0:
            b = a.copy()
        elif kind == 7:
            b = rs.randint(0, nsym, max(1, length + rs.randint(-300, 301))).astype(np.uint8)
        else:
            p = [0.002, 0.005, 0.01, 0.02, 0.04, 0.1][kind - 1]
            keep = rs.random_sample(length) >= p
            ins = rs.random_sample(length) < p
            out = []
            for j in range(length):
                if ins[j]:
                    out.append(rs.randint(0, nsym))
                if keep[j]:
                    out.append(a[j])
            b = np.array(out, dtype=np.uint8)
            if kind == 3:                            # one long gap on top: length difference beyond small windows
                cut = rs.randint(0, len(b) // 2)
                b = np.concatenate([b[:cut], b[cut + rs.randint(20, 140):]])
        if i % 3 == 0:
            a, b = b, a
        seqs.append((a, b))
    syms = np.concatenate([x for ab in seqs for x in ab] + [np.zeros(1, np.uint8)])
    a_off, a_len, b_off, b_len = [], [], [], []
    pos = 0
    for a, b in seqs:
        a_off.append(pos); a_len.append(len(a)); pos += len(a)
        b_off.append(pos); b_len.append(len(b)); pos += len(b)
    return (syms, np.array(a_off, np.uint64), np.array(a_len, np.uint32), np.array(b_off, np.uint64),
            np.array(b_len, np.uint32))


@pytest.mark.parametrize("nlw,nsym", [(3, 4), (2, 4), (4, 3), (5, 4), (8, 2), (3, 7), (4, 13), (0, 4)])
def test_edit_distance_window_class(native, nlw, nsym, monkeypatch):
    """The window class (malcs_diag.hip) for every window size, with pairs it settles, pairs it hands
    back to the full-matrix kernel and pairs that never enter it; nlw = 0 switches the class off."""
    from oracle import oracle as O
    monkeypatch.setenv("MALCS_DIAG_NLW", str(nlw))
    data = _window_class_batch(np.random.RandomState(100 + nlw + nsym), 1500, nsym)
    batch = native.Batch(native.default_context(), *data, n_symbols=nsym)
    info = batch.plan_info()
    assert (info["window_pairs"] > 0) == (nlw > 0)
    assert info["window_limbs"] == (nlw if nlw else 0)
    got = batch.distances()
    again = batch.distances()                      # second launch: the retry counters start over
    if nlw:                                        # the measurement entry (an event pair per launch) leaves the results alone
        each = batch.kernel_durations(3)
        assert each.shape == (3,) and (each > 0).all() and (each < 50).all()
        third = batch.distances()
    else:
        with pytest.raises(native.MalcsError):
            batch.kernel_durations(3)
        third = again
    batch.close()
    ref = O.edit_distance_batch(*data)
    assert np.array_equal(got, ref)
    assert np.array_equal(again, ref)
    assert np.array_equal(third, ref)
    if nlw:
        assert (ref > 32 * nlw - 32).sum() > 50 and (ref <= 32 * nlw - 32).sum() > 50


@pytest.mark.parametrize("nsym", [4, 7, 13])
def test_edit_distance_window_class_mixed_tiles(native, nsym):
    """Tiles of the default window class whose lanes differ in length, length difference and divergence: the
    three-block statement of the plain stretch, the masked first / last blocks and the retry list in one batch."""
    from oracle import oracle as O
    data = gen.window_mix(6000, seed=11 + nsym, nsym=nsym)
    batch = native.Batch(native.default_context(), *data, n_symbols=nsym)
    info = batch.plan_info()
    got = batch.distances()
    batch.close()
    ref = O.edit_distance_batch(*data)
    assert info["window_pairs"] > 5000 and info["window_limbs"] == 3
    assert (ref > info["window_threshold"]).sum() > 500
    assert np.array_equal(got, ref)


@pytest.mark.parametrize("lds", ["1", "0"])
def test_edit_distance_two_limb_pieces(native, lds, monkeypatch):
    """Two-limb pieces of every length and both parities (whole tiles, front and back parts), with the match masks
    from LDS (generated stream, malcs_diag_lds.inc) and folded from the planes: same distances as the oracle."""
    from oracle import oracle as O
    monkeypatch.setenv("MALCS_DIAG_HALF", "1")
    monkeypatch.setenv("MALCS_DIAG_LDS", lds)
    for cut in ("1", "0"):
        monkeypatch.setenv("MALCS_DIAG_CUT", cut)
        for length, p, seed in [(70, 0.02, 31), (130, 0.03, 32), (200, 0.02, 33), (330, 0.03, 34), (460, 0.02, 35), (1500, 0.01, 36)]:
            data = gen.c2_batch(700, seed=seed, length=length, p=p)
            batch = native.Batch(native.default_context(), *data, n_symbols=4)
            info = batch.plan_info()
            got = batch.distances()
            again = batch.distances()
            batch.close()
            ref = O.edit_distance_batch(*data)
            assert info["two_limb_tiles"] == info["window_tiles"]
            assert info["window_tiles"] > 0 or length < 330
            assert np.array_equal(got, ref) and np.array_equal(again, ref), (cut, length)


@pytest.mark.parametrize("half", ["0", "1", ""])
def test_edit_distance_two_limb_mode(native, half, monkeypatch):
    """The two-limb mode of the window class (threshold 48): forced on, forced off and chosen from a sample of the
    batch, on batches whose distances stay below the threshold, straddle it and exceed it."""
    from oracle import oracle as O
    if half:
        monkeypatch.setenv("MALCS_DIAG_HALF", half)
    else:
        monkeypatch.delenv("MALCS_DIAG_HALF", raising=False)
    for pairs, length, p, seed in [(6000, 1000, 0.01, 21), (6000, 1000, 0.02, 22), (5000, 700, 0.04, 23), (5000, 600, 0.0, 24)]:
        data = gen.c2_batch(pairs, seed=seed, length=length, p=p)
        batch = native.Batch(native.default_context(), *data, n_symbols=4)
        info = batch.plan_info()
        got = batch.distances()
        again = batch.distances()
        batch.close()
        ref = O.edit_distance_batch(*data)
        assert np.array_equal(got, ref) and np.array_equal(again, ref)
        assert info["window_tiles"] > 0
        if half == "1":
            assert info["two_limb_tiles"] == info["window_tiles"]
        elif half == "0":
            assert info["two_limb_tiles"] == 0
        else:                                       # the sample decides: similar batches only
            assert (info["two_limb_tiles"] > 0) == (p <= 0.01), (p, int(ref.max()), info)
    data = gen.window_mix(5000, seed=12)
    batch = native.Batch(native.default_context(), *data, n_symbols=4)
    got = batch.distances()
    batch.close()
    assert np.array_equal(got, O.edit_distance_batch(*data))


@pytest.mark.parametrize("lds3", ["", "0", "2"])
def test_three_limb_pieces_lds_stream_and_plane_paths(native, lds3, monkeypatch):
    """Three-limb pieces of DNA batches run through a generated stream that reads its match masks from LDS
    (DIAG_LDS3_STREAM; their text rows hold 2-bit symbols).  MALCS_DIAG_LDS3=0 sends the same rows through the
    plane-folding paths, which turn the symbols back into planes, =2 leaves the stream half way through every piece and
    hands the rest to them; MALCS_DIAG_LDS=0 (no symbols in the rows at all) is
    the stream of round 2.  Forced three limbs, cut and uncut tiles, lengths whose pieces are a few blocks and many:
    all three give the oracle's distances."""
    from oracle import oracle as O
    monkeypatch.setenv("MALCS_DIAG_HALF", "0")
    if lds3:
        monkeypatch.setenv("MALCS_DIAG_LDS3", lds3)
    for pairs, length, p, cut, seed in [(6000, 1000, 0.02, "", 41), (5000, 330, 0.03, "1", 42), (4200, 2500, 0.01, "1", 43), (9000, 400, 0.05, "0", 44)]:
        if cut:
            monkeypatch.setenv("MALCS_DIAG_CUT", cut)
        else:
            monkeypatch.delenv("MALCS_DIAG_CUT", raising=False)
        data = gen.c2_batch(pairs, seed=seed, length=length, p=p)
        ref = O.edit_distance_batch(*data)
        batch = native.Batch(native.default_context(), *data, n_symbols=4)
        info = batch.plan_info()
        got = batch.distances()
        batch.close()
        assert info["window_tiles"] > 0 and info["two_limb_tiles"] == 0
        assert np.array_equal(got, ref), (length, p, cut)
    monkeypatch.setenv("MALCS_DIAG_LDS", "0")
    data = gen.c2_batch(5000, seed=45, length=800, p=0.02)
    batch = native.Batch(native.default_context(), *data, n_symbols=4)
    got = batch.distances()
    batch.close()
    assert np.array_equal(got, O.edit_distance_batch(*data))


@pytest.mark.parametrize("pad", ["0", "1", "1000"])
def test_window_kernel_with_other_numbers_of_leading_workgroups(native, pad, monkeypatch):
    """The window kernel's launch starts with workgroups that return at once (one per CU: they take the hardware's
    first turn so that the pieces land where the deal expects them, malcs_diag_launch).  Their number moves the pieces'
    workgroup indices, nothing else: none, one, and a number that is no multiple of anything give the same distances,
    first launch and replays, cut tiles included."""
    from oracle import oracle as O
    monkeypatch.setenv("MALCS_DIAG_PAD", pad)
    for pairs, length, p, seed in [(20000, 1000, 0.01, 51), (5000, 600, 0.03, 52)]:
        data = gen.c2_batch(pairs, seed=seed, length=length, p=p)
        batch = native.Batch(native.default_context(), *data, n_symbols=4)
        got = batch.distances()
        batch.replan()
        batch.launch_distances()
        again = batch.fetch_distances()
        batch.close()
        ref = O.edit_distance_batch(*data)
        assert np.array_equal(got, ref) and np.array_equal(again, ref), (pad, length)


def test_batches_in_flight_on_streams_of_their_own(native):
    """bench.py keeps three batches in flight, each on a context (= stream) of its own, so that one batch's plan runs
    beside another's distance kernels.  Plans and launches queued in turn, many times over, without a wait in between:
    every batch's distances are the oracle's, whatever the others were doing on the card at the time."""
    from oracle import oracle as O
    ctxs = [native.Context(0) for _ in range(3)]
    datas = [gen.c2_batch(40000, seed=90 + k, length=(1000, 700, 1300)[k], p=(0.01, 0.02, 0.005)[k]) for k in range(3)]
    batches = [native.Batch(c, *d, n_symbols=4) for c, d in zip(ctxs, datas)]
    for i in range(60):
        b = batches[i % 3]
        b.replan()
        b.launch_distances()
    got = [b.fetch_distances() for b in batches]
    for b in batches:
        b.close()
    for c in ctxs:
        c.close()
    for d, g in zip(datas, got):
        assert np.array_equal(g, O.edit_distance_batch(*d))


@pytest.mark.parametrize("wave", ["", "1"])
def test_sample_estimate_bounds_the_distances(native, wave, monkeypatch):
    """The sample that picks the window class's mode (malcs_plan.inc: one lane per part of a pair, cuts probed on the
    device; MALCS_SAMPLE_WAVE=1: the full-matrix wave over sub-pairs the host cut, which sequences beyond 4 096 symbols
    always get) estimates a pair's distance from above.  Its mean must lie between the batch's own mean (less the
    sampling error of 1 000 pairs) and a third more; the distances themselves are the oracle's in every case."""
    from oracle import oracle as O
    if wave:
        monkeypatch.setenv("MALCS_SAMPLE_WAVE", wave)
    for pairs, length, p, seed in [(8000, 330, 0.01, 31), (8000, 1000, 0.01, 32), (6000, 1900, 0.005, 33), (5000, 4000, 0.002, 34),
                                   (4200, 6000, 0.001, 35), (8000, 700, 0.03, 36)]:
        data = gen.c2_batch(pairs, seed=seed, length=length, p=p)
        batch = native.Batch(native.default_context(), *data, n_symbols=4)
        info = batch.plan_info()
        got = batch.distances()
        batch.close()
        ref = O.edit_distance_batch(*data)
        assert np.array_equal(got, ref), (length, p)
        if info["window_tiles"] == 0:
            continue
        mean, est = float(ref.mean()), info["sample_mean_x16"] / 16.0
        assert mean - 4.0 * float(ref.std()) / 30.0 - 0.1 <= est <= 1.34 * mean + 2.0, (length, p, mean, est, info)


@pytest.mark.parametrize("pairs,length,p,half", [(100000, 1000, 0.01, ""), (6000, 1000, 0.01, ""), (5000, 700, 0.04, ""), (7000, 460, 0.02, "1"),
                                                 (3000, 1500, 0.01, "1"), (300000, 330, 0.02, "")])
def test_window_plan_made_on_the_device(native, pairs, length, p, half, monkeypatch):
    """The plan of the window class is made on the device and never comes back to the host for a launch (malcs_plan.inc);
    read back through malcs_batch_plan_dump it must hold together: every pair of the class in one slot, slots sorted by
    (group, blocks), the pieces of a tile cover its blocks, rows of different pieces do not overlap, the table of the
    launch names every piece once -- after the first plan and after plans queued again behind launches -- and the
    distances are the oracle's.  Sizes: all pieces dealt in LDS (cut, balanced and not), uncut three-limb tiles, more
    tiles than the deal takes (the longest-first order)."""
    from oracle import oracle as O
    from util import check_window_plan
    if half:
        monkeypatch.setenv("MALCS_DIAG_HALF", half)
    data = gen.c2_batch(pairs, seed=77, length=length, p=p)
    batch = native.Batch(native.default_context(), *data, n_symbols=4)
    first = check_window_plan(batch)
    ref = O.edit_distance_batch(*data) if pairs <= 100000 else None
    d0 = batch.distances()
    for _ in range(3):
        batch.replan()
        batch.launch_distances()
    again = check_window_plan(batch)
    d1 = batch.fetch_distances()
    batch.close()
    assert first["pieces"] == again["pieces"] and first["rows"] == again["rows"]
    assert np.array_equal(d0, d1)
    if ref is not None:
        assert np.array_equal(d0, ref)
    else:
        pick = np.random.RandomState(1).choice(pairs, 2000, replace=False)
        syms, a_off, a_len, b_off, b_len = data
        assert np.array_equal(d0[pick], O.edit_distance_batch(syms, a_off[pick], a_len[pick], b_off[pick], b_len[pick]))
    if first.get("pieces", 0) <= 4096 and pairs >= 6000:
        assert first["simd_load_max"] == first["reported_worst"]


def _as_packed(native, args, n_planes, rs=None, on_device=False):
    """The batch of byte sequences `args` = (syms, a_off, a_len, b_off, b_len) again through malcs_batch_create_packed, from
    an arena this test packs with numpy (util.pack_arena: include/malcs.h's format restated)."""
    from util import pack_arena
    syms, a_off, a_len, b_off, b_len = args
    n = len(a_len)
    planes, where = pack_arena(syms, np.concatenate([a_off, b_off]), np.concatenate([a_len, b_len]), n_planes, junk_tails=rs)
    ctx = native.default_context()
    if not on_device:
        return native.Batch.from_packed(ctx, planes, where[:n], a_len, where[n:], b_len, n_planes)
    from util import hip_runtime, to_device
    resident = to_device(planes)
    try:
        batch = native.Batch.from_packed(ctx, resident.value, where[:n], a_len, where[n:], b_len, n_planes, n_dwords=planes.size)
        ctx.sync()                                                     # the library has its copy: the caller's arena may go
    finally:
        hip_runtime().hipFree(resident)
    return batch


@pytest.mark.parametrize("n_symbols,n_planes", [(4, 2), (7, 3), (13, 4), (200, 8)])
def test_packed_batches_word_boundaries(native, n_symbols, n_planes):
    """malcs_batch_create_packed: the caller's bit planes instead of bytes, junk behind every sequence's last symbol."""
    from oracle import oracle as O
    rs = np.random.RandomState(700 + n_symbols)
    lengths = [0, 1, 31, 32, 33, 63, 64, 65, 127, 128, 129, 255, 256, 257, 511, 512, 513, 1000, 1023, 1024, 1025, 2049, 4097]
    flat = []
    for m in lengths:
        a = rs.randint(0, n_symbols, m).astype(np.uint8)
        keep = rs.random_sample(m) > 0.02
        flat += [a, np.concatenate([a[keep], rs.randint(0, n_symbols, rs.randint(0, 4)).astype(np.uint8)])]
        flat += [a, rs.randint(0, n_symbols, max(0, m + rs.randint(-2, 3))).astype(np.uint8)]
    off = np.concatenate([[0], np.cumsum([len(x) for x in flat])]).astype(np.uint64)
    lens = np.array([len(x) for x in flat], np.uint32)
    syms = np.concatenate(flat + [np.zeros(1, np.uint8)])
    args = (syms, off[0:-1:2], lens[0::2], off[1::2], lens[1::2])
    want = O.edit_distance_batch(*args).tolist()
    for on_device in (False, True):
        batch = _as_packed(native, args, n_planes, rs, on_device)
        assert batch.distances().tolist() == want
        batch.replan()
        batch.launch_distances()
        assert batch.fetch_distances().tolist() == want
        batch.close()


def test_packed_batch_c2_and_graphs(native):
    """The C2 workload from planes that are in HBM already: distances, plan invariants and LCS graphs are those of the
    batch made from bytes; pairs that share a sequence share its blocks."""
    from util import check_window_plan
    data = gen.c2_batch(20000, seed=9)
    ctx = native.default_context()
    plain = native.Batch(ctx, *data, 4)
    want = plain.distances()
    packed = _as_packed(native, data, 2, np.random.RandomState(1), on_device=True)
    assert np.array_equal(packed.distances(), want)
    assert packed.plan_info() == plain.plan_info()
    check_window_plan(packed)
    plain.close(); packed.close()
    # graphs, and pairs that share their reference window (one run of blocks, many pairs)
    from util import pack_arena
    small = gen.c2_batch(64, seed=10)
    syms, a_off, a_len, b_off, b_len = small
    planes, where = pack_arena(syms, np.concatenate([a_off[:1], b_off]), np.concatenate([a_len[:1], b_len]), 2)
    shared = native.Batch.from_packed(ctx, planes, np.repeat(where[:1], 64), np.repeat(a_len[:1], 64), where[1:], b_len, 2)
    plain = native.Batch(ctx, syms, np.repeat(a_off[:1], 64), np.repeat(a_len[:1], 64), b_off, b_len, 4)
    assert np.array_equal(shared.distances(), plain.distances())
    g1, g2 = shared.graphs(), plain.graphs()
    assert np.array_equal(g1.counts, g2.counts)
    for i in range(0, 64, 7):
        assert np.array_equal(g1.nodes(i), g2.nodes(i)) and np.array_equal(g1.edges(i), g2.edges(i))
    g1.close(); g2.close(); shared.close(); plain.close()


def test_packed_batch_bad_arguments(native):
    ctx = native.default_context()
    planes = np.zeros(8, np.uint32)
    with pytest.raises(native.MalcsError, match="n_planes"):
        native.Batch.from_packed(ctx, planes, [0], [10], [2], [10], 5)
    with pytest.raises(native.MalcsError, match="whole blocks inside the arena"):
        native.Batch.from_packed(ctx, planes, [0], [10], [1], [10], 2)        # not at a block
    with pytest.raises(native.MalcsError, match="whole blocks inside the arena"):
        native.Batch.from_packed(ctx, planes, [0], [10], [6], [40], 2)        # runs past the end
    ok = native.Batch.from_packed(ctx, planes, [0], [64], [4], [64], 2)       # 2 + 2 blocks: the arena exactly
    assert ok.distances().tolist() == [0]
    ok.close()
    empty = native.Batch.from_packed(ctx, np.zeros(0, np.uint32), [], [], [], [], 2)
    assert empty.distances().tolist() == []
    empty.close()
    lone = native.Batch.from_packed(ctx, planes, [0], [0], [0], [33], 2)      # an empty sequence against 33 symbols
    assert lone.distances().tolist() == [33]
    lone.close()


def test_empty_batch_and_bad_arguments(native):
    import algebra_amd
    assert algebra_amd.edit_distance_batch([]).tolist() == []
    assert algebra_amd.edit_distance("", "") == 0
    assert algebra_amd.edit_distance("", "ACGT") == 4
    ctx = native.default_context()
    with pytest.raises(native.MalcsError):
        native.Batch(ctx, np.zeros(4, np.uint8), [0], [10], [0], [2], 4)     # sequence outside the buffer
    with pytest.raises(native.MalcsError):
        native.Batch(ctx, np.zeros(4, np.uint8), [0], [2], [0], [2], 0)      # n_symbols out of range
    # empty batches and NULL handles through every family
    import ctypes
    empty = native.Batch(ctx, np.zeros(1, np.uint8), [], [], [], [], 4)
    assert empty.distances().tolist() == []
    for res in (empty.graphs(), empty.lcs_nodes()):
        assert res.counts.shape == (0, 4)
        res.close()
    empty.close()
    handle = ctypes.c_void_p()
    assert native.lib.malcs_lcs_graph_batch(None, None, None, ctypes.byref(handle)) != 0
    assert native.lib.malcs_lcs_nodes_batch(None, None, None, ctypes.byref(handle)) != 0
    assert native.lib.malcs_relmatrix_create(None, 0, None, None, None, None, 16, 0, 1, ctypes.byref(handle)) != 0
    assert b"argument" in native.lib.malcs_last_error()
    none = native.RelationMatrix(ctx, [], [], [], [], 16)
    assert none.pending().shape == (0, 2) and none.assemble().shape == (0, 0)
    none.close()
    with pytest.raises(native.MalcsError):
        native.RelationMatrix(ctx, [0], [1], [0], [1], 16, rank=2, world=2)   # rank outside the world


# ---------------------------------------------------------------------------------------------
# family 2: graphs
# ---------------------------------------------------------------------------------------------

def build(cases):
    from algebra_amd.lcs import LCSgraph
    return LCSgraph.batch([(c["reference"], c["observed"]) for c in cases], [c["shift"] for c in cases])


def test_graph_reference_vectors(native):
    graphs = build(REF["lcs"])
    for rec, graph in zip(REF["lcs"], graphs):
        assert_graph(rec, graph_record(graph))
        obs = observables(graph)
        if "observables" in rec:
            assert obs == rec["observables"], (rec["reference"], rec["observed"])
        else:
            assert {k: digest(v) for k, v in obs.items()} == rec["observables_sha256"]


def test_graph_like_the_reference_tests(native):
    """A few vectors written the way tests/lcs/test_lcs_graph.py writes them."""
    from algebra_amd.lcs import LCSgraph
    from algebra_amd.variants import Variant
    assert LCSgraph("", "").edges() == set()
    assert LCSgraph("A", "A").edges() == set()
    assert LCSgraph("TCTC", "TC").edges() == {Variant(0, 2, ""), Variant(1, 3, ""), Variant(2, 4, "")}
    assert list(LCSgraph("AAAA", "AAA").nodes()) == [LCSgraph.Node(0, 0, 3), LCSgraph.Node(1, 0, 3)]
    assert list(LCSgraph("GACTACGAGACAC", "TACGAGACAC").nodes()) == [LCSgraph.Node(0, 0, 0), LCSgraph.Node(3, 0, 0)]
    graph = LCSgraph("TTGGTT", "AAGTTAG")
    assert [e for *_, e in graph.bfs_traversal()] == [
        [Variant(0, 0, "AAG")], [Variant(0, 2, "AA")], [Variant(0, 3, "AA")], [Variant(2, 2, "A")],
        [Variant(2, 3, "A")], [Variant(3, 4, "")], [Variant(6, 6, "AG")], [Variant(3, 6, "")], [Variant(4, 6, "")]]
    assert list(LCSgraph("TTT", "TATTTT").paths())[:3] == [
        [Variant(1, 1, "A"), Variant(3, 3, "TT")],
        [Variant(1, 1, "A"), Variant(2, 2, "T"), Variant(3, 3, "T")],
        [Variant(1, 1, "A"), Variant(2, 2, "TT")]]
    assert list(LCSgraph("", "").paths()) == [[]]
    assert str(LCSgraph.Node(0, 0, 0)) == "(0, 0, 0)"


def test_graph_random_small(native):
    graphs = build(SMALL["lcs"])
    for rec, graph in zip(SMALL["lcs"], graphs):
        assert_graph(rec, graph_record(graph))


def test_graph_random_medium(native):
    graphs = build(MEDIUM["lcs"])
    for rec, graph in zip(MEDIUM["lcs"], graphs):
        got = graph_record(graph)
        assert got["distance"] == rec["distance"] and got["supremal"] == rec["supremal"], rec["name"]
        assert len(got["nodes"]) == rec["n_nodes"] and len(got["edges"]) == rec["n_edges"], rec["name"]
        assert digest(got) == rec["graph_sha256"], rec["name"]


@pytest.mark.parametrize("path", ["sweep", "sweep-all", "band"])
def test_graph_arrays_vs_oracle(native, path, monkeypatch):
    """Straight through the C-ABI (no facade objects): arrays equal the oracle's, runs are a
    sub-list of the oracle's level lists.  Windows below 4 096 rows: bit-vector kernels ("band" = sweeps off) and
    one-wave sweeps (by default from 192 rows and distance <= length / 8 on, "sweep-all" from 1 row on at any distance <= 510)."""
    from oracle import oracle as O
    from algebra_amd.lcs.encoding import encode_pairs
    use_path(monkeypatch, path)
    rs = np.random.RandomState(77)
    pairs, shifts = [], []
    for _ in range(300):
        kind = ["random", "indel", "sparse", "tandem"][rs.randint(0, 4)]
        a, b = gen.pair_of_kind(rs, kind, int(rs.randint(1, 400)), "ACGT"[:rs.randint(1, 5)])
        pairs.append((a, b))
        shifts.append(int(rs.randint(0, 50)))
    for m in (191, 192, 193, 1000, 2047, 2048, 4095):
        a = gen.random_seq(rs, m)
        for p in (0.0, 0.002, 0.02, 0.06):
            pairs.append((a, gen.mutate(rs, a, p, p)))
            shifts.append(0)
        pairs.append((a, a[m // 3:]))
        shifts.append(7)
    syms, a_off, a_len, b_off, b_len, n_symbols = encode_pairs(pairs)
    batch = native.Batch(native.default_context(), syms, a_off, a_len, b_off, b_len, n_symbols)
    res = batch.graphs(np.array(shifts, np.int64))
    batch.close()
    for i, (a, b) in enumerate(pairs):
        ref = O.lcs_graph(a, b, shifts[i])
        assert res.distance[i] == ref.distance
        assert tuple(res.supremal[i]) == ref.supremal
        assert np.array_equal(res.nodes(i), ref.nodes) and np.array_equal(res.edges(i), ref.edges)
        mine = [tuple(r) for r in res.runs(i).tolist()]
        theirs = [tuple(r) for r in ref.runs[np.argsort(ref.runs[:, 3], kind="stable")].tolist()]
        it = iter(theirs)
        assert all(r in it for r in mine), (a, b)          # ordered sub-list
    res.close()


def test_lcs_nodes_facade(native):
    """_lcs_nodes as the reference's test reads it (tests/lcs/test_lcs_graph.py:20-51): exact ordered
    level lists, off-path runs and empty levels included."""
    from algebra_amd.lcs import LCSgraph
    from algebra_amd.lcs.lcs_graph import _lcs_nodes
    assert _lcs_nodes("", "") == (0, [])
    assert _lcs_nodes("A", "C") == (2, [[]])
    assert _lcs_nodes("ACGT", "") == (4, [])
    assert _lcs_nodes("AA", "ACA") == (1, [[LCSgraph.Node(0, 0, 1)], [LCSgraph.Node(1, 2, 1)]])
    for rec in REF["lcs"][::3]:
        a, b = rec["reference"], rec["observed"]
        distance, levels = _lcs_nodes(a, b, rec["shift"])
        assert distance == rec["edit_distance"]
        got = [[[n.row, n.col, n.length] for n in level] for level in levels]
        if "levels" in rec:
            assert got == rec["levels"], (a, b)
        else:
            assert digest(got) == rec["levels_sha256"]
        if levels and any(levels):
            assert distance == len(a) - len(levels) + len(b) - len(levels)      # reference test :50


def lcs_nodes_batch(native, cases):
    from algebra_amd.lcs.encoding import encode_pairs
    syms, a_off, a_len, b_off, b_len, n_symbols = encode_pairs([(c["reference"], c["observed"]) for c in cases])
    batch = native.Batch(native.default_context(), syms, a_off, a_len, b_off, b_len, n_symbols)
    res = batch.lcs_nodes(np.array([c["shift"] for c in cases], np.int64))
    batch.close()
    out = []
    for i in range(len(cases)):
        levels = [[] for _ in range(int(res.counts[i, 1]))]
        for row, col, length, level in res.runs(i).tolist():
            levels[level].append([row, col, length])
        out.append((int(res.distance[i]), levels))
    assert not res.counts[:, 2:].any()                 # runs only: no graph in this handle
    res.close()
    return out


def test_lcs_nodes_exact_level_lists(native):
    """malcs_lcs_nodes_batch against every level list captured from the reference (880 pairs)."""
    cases = REF["lcs"] + SMALL["lcs"] + MEDIUM["lcs"]
    for rec, (distance, levels) in zip(cases, lcs_nodes_batch(native, cases)):
        assert distance == rec["edit_distance"]
        if "levels" in rec:
            assert levels == rec["levels"], (rec["reference"], rec["observed"], rec["shift"])
        else:
            assert len(levels) == rec["n_levels"] and digest(levels) == rec["levels_sha256"], rec.get("name")


def test_lcs_nodes_vs_oracle(native):
    """Seeded pairs beyond the fixtures' sizes (word and block boundaries, 1..9 symbols, divergent pairs whose
    diagonals do not fit LDS) against the oracle's restatement of the sweep."""
    from oracle import oracle as O
    rs = np.random.RandomState(4242)
    cases = []
    for _ in range(160):
        kind = ["random", "indel", "sparse", "tandem"][rs.randint(0, 4)]
        a, b = gen.pair_of_kind(rs, kind, int(rs.randint(1, 700)), "ACGTNRYKM"[:rs.randint(1, 10)])
        cases.append({"reference": a, "observed": b, "shift": int(rs.randint(0, 9))})
    for m in (31, 32, 33, 63, 64, 65, 1000, 1024, 1025):
        a, b = gen.pair_of_kind(rs, "indel", m, "ACGT")
        cases.append({"reference": a, "observed": b, "shift": 0})
        cases.append({"reference": a, "observed": a[: m // 2], "shift": 3})
    a = "".join(rs.choice(list("AC"), 9000))
    b = "".join(rs.choice(list("GT"), 9100))
    cases.append({"reference": a, "observed": b, "shift": 0})              # 18 100 diagonals: FR in HBM, no matches
    a, b = gen.pair_of_kind(rs, "random", 3000, "ACGT")
    cases.append({"reference": a, "observed": b, "shift": 0})              # random 3 kb pair, ~2 000 diagonals, ~1e6 runs
    for c, (distance, levels) in zip(cases, lcs_nodes_batch(native, cases)):
        ref = O.lcs_graph(c["reference"], c["observed"], c["shift"])
        assert distance == ref.distance
        assert levels == [[list(t) for t in lvl] for lvl in ref.level_lists()], (len(c["reference"]), len(c["observed"]))


def test_max_distance(native):
    from algebra_amd.lcs import LCSgraph
    from algebra_amd.lcs.lcs_graph import _lcs_nodes
    for rec in REF["max_distance"] + SMALL["max_distance"]:
        a, b, md = rec["reference"], rec["observed"], rec["max_distance"]
        if "error" in rec:
            with pytest.raises(ValueError) as exc:
                _lcs_nodes(a, b, max_distance=md)
            assert str(exc.value) == rec["error"]
            with pytest.raises(ValueError):
                LCSgraph(a, b, max_distance=md)
        else:
            assert _lcs_nodes(a, b, max_distance=md)[0] == rec["distance"]
            assert LCSgraph(a, b, max_distance=md).distance == rec["distance"]


def test_max_distance_per_pair_status(native):
    """All the reference's max_distance cases as ONE batch: raise_on_limit=False reports the pairs beyond their limit
    one by one (None + status) instead of losing the batch to the first of them."""
    from algebra_amd.lcs import LCSgraph
    recs = REF["max_distance"] + SMALL["max_distance"]
    assert any("error" in r for r in recs) and any("error" not in r for r in recs)
    pairs = [(r["reference"], r["observed"]) for r in recs]
    limits = [r["max_distance"] for r in recs]
    with pytest.raises(ValueError, match="maximum distance exceeded"):
        LCSgraph.batch(pairs, max_distances=limits)
    graphs = LCSgraph.batch(pairs, max_distances=limits, raise_on_limit=False)
    assert len(graphs) == len(recs)
    for rec, graph, status in zip(recs, graphs, graphs.status.tolist()):
        if "error" in rec:
            assert graph is None and status == native.PAIR_MAX_DISTANCE_EXCEEDED
        else:
            assert status == native.PAIR_OK and graph.distance == rec["distance"]
            assert graph.supremal == LCSgraph(rec["reference"], rec["observed"]).supremal
    assert LCSgraph.batch([], raise_on_limit=False).status.tolist() == []


def test_from_sequence(native):
    from algebra_amd.lcs import LCSgraph
    for rec in REF["sequences"] + SEQS["sequences"]:
        graph = LCSgraph.from_sequence(rec["reference"], rec["observed"])
        assert_graph(rec, graph_record(graph))
        if "observables" in rec:
            assert observables(graph) == rec["observables"]


def test_from_variant_and_from_supremal(native):
    from algebra_amd.lcs import LCSgraph
    from algebra_amd.variants import Variant
    for rec in REF["from_variant"]:
        variants = [Variant(*v) for v in rec["variants"]]
        graph = LCSgraph.from_variant(rec["reference"], variants, rec["offset"])
        assert graph_record(graph) == rec["graph"]
    for rec in REF["from_supremal"]:
        graph = LCSgraph.from_supremal(rec["reference"], Variant(*rec["supremal"]))
        assert graph_record(graph) == rec["graph"]


@pytest.mark.parametrize("path", ["sweep", "band"])
@pytest.mark.parametrize("idx", range(len(LARGE["from_sequence"])))
def test_large_from_sequence(native, idx, path, monkeypatch):
    """BASELINE configs 3 and 5 (i): 10 kb tandem-repeat windows, 100 kb and 1 Mb sparse pairs; full graph,
    SHA-256 of the canonical record captured from the reference."""
    from algebra_amd.lcs import LCSgraph
    use_path(monkeypatch, path)
    rec = LARGE["from_sequence"][idx]
    a, b = gen.large_pair(rec["case"])
    assert digest([a, b]) == rec["input_sha256"]
    graph = LCSgraph.from_sequence(a, b)
    got = graph_record(graph)
    assert got["distance"] == rec["distance"]
    assert [got["supremal"][0], got["supremal"][1]] == rec["supremal_pos"]
    assert len(got["nodes"]) == rec["n_nodes"] and len(got["edges"]) == rec["n_edges"]
    assert digest(got) == rec["graph_sha256"]


@pytest.mark.parametrize("path", ["sweep", "band"])
def test_c3_and_c5_i_at_the_config_count(native, path, monkeypatch):
    """BASELINE config 3 at its own count (16 pairs of 10 kb tandem-repeat windows, the generator and seeds bench.py
    times) and config 5 (i) at its (4 sparse 1 Mb pairs), as ONE batch each, on both device paths.  The reference's own
    outputs pin two of the C3 pairs and one of the 1 Mb pairs (test_large_from_sequence); the others are checked against
    the C oracle, which those fixtures pin in turn: parity with the reference itself is unpinned for them."""
    from oracle import oracle as O
    from algebra_amd.lcs import LCSgraph
    use_path(monkeypatch, path)
    c3 = [gen.tandem_pair(np.random.RandomState(300 + s), 10000, 20) for s in range(16)]
    c5 = [gen.large_pair({"kind": "few", "seed": 50 + i, "length": 1000000, "events": 20}) for i in range(4)]
    for pairs in (c3, c5):
        graphs = LCSgraph.batch(pairs)
        for (a, b), graph in zip(pairs, graphs):
            ref = O.lcs_graph(a, b)
            assert graph.distance == ref.distance
            assert (graph.supremal.start, graph.supremal.end) == tuple(ref.supremal[:2])
            result, index = graph._result
            assert np.array_equal(result.nodes(index), ref.nodes) and np.array_equal(result.edges(index), ref.edges)


def test_graph_size_limit_is_loud(native):
    from algebra_amd.lcs import LCSgraph
    rs = np.random.RandomState(9)
    a, b = gen.random_seq(rs, 60000), gen.random_seq(rs, 60000)      # ~42 000 edits: wider than any band
    with pytest.raises(native.MalcsError, match="exceed"):
        LCSgraph(a, b)


@pytest.mark.parametrize("path", ["sweep", "band", "mixed"])
def test_graph_band_vs_oracle(native, path, monkeypatch):
    """Windows beyond 4 095 symbols: the two sweeps, or the sliding-window kernels at all band widths; both signs
    of n - m."""
    from oracle import oracle as O
    from algebra_amd.lcs.encoding import encode_pairs
    use_path(monkeypatch, path)
    rs = np.random.RandomState(21)
    pairs = []
    for m, p in ((4096, 0.002), (5000, 0.01), (8000, 0.05), (12000, 0.1), (9000, 0.3), (20000, 0.004)):
        a = gen.random_seq(rs, m)
        pairs.append((a, gen.mutate(rs, a, p, p)))
    a = gen.random_seq(rs, 7000)
    pairs += [(a, a[:3000] + a[3500:]), (a[:2000] + a[2600:], a), (a, a), tuple(gen.tandem_pair(rs, 9000, 12))]
    syms, a_off, a_len, b_off, b_len, n_symbols = encode_pairs(pairs)
    batch = native.Batch(native.default_context(), syms, a_off, a_len, b_off, b_len, n_symbols)
    res = batch.graphs(np.arange(len(pairs), dtype=np.int64))
    batch.close()
    for i, (a, b) in enumerate(pairs):
        ref = O.lcs_graph(a, b, i)
        assert res.distance[i] == ref.distance and tuple(res.supremal[i]) == ref.supremal, i
        assert np.array_equal(res.nodes(i), ref.nodes) and np.array_equal(res.edges(i), ref.edges), i
    res.close()


@pytest.mark.parametrize("path", ["sweep", "band"])
@pytest.mark.parametrize("alphabet", ["AC", "ACG", "ACGTN", "ACGTNRYKMSWBDHV", "".join(chr(65 + i) for i in range(26)) + "0123456789"])
def test_graph_long_windows_any_alphabet(native, alphabet, path, monkeypatch):
    """Long windows over 2 ... 36 symbols (1, 2, 3, 4 and 8 bit planes), lengths off the word grid, reference
    longer / shorter than the observed sequence, a pure deletion and a pure insertion; graphs against the oracle."""
    from oracle import oracle as O
    from algebra_amd.lcs.encoding import encode_pairs
    use_path(monkeypatch, path)
    rs = np.random.RandomState(len(alphabet))
    pairs = []
    for m, p in ((4099, 0.003), (6001, 0.02), (5055, 0.0005)):
        a = gen.random_seq(rs, m, alphabet)
        pairs.append((a, gen.mutate(rs, a, p, p, alphabet)))
    a = gen.random_seq(rs, 4500, alphabet)
    pairs += [(a, a[:1000] + a[1033:]), (a[:4100] + a[4131:], a), (a[17:], a), (a, a[:-1])]
    syms, a_off, a_len, b_off, b_len, n_symbols = encode_pairs(pairs)
    batch = native.Batch(native.default_context(), syms, a_off, a_len, b_off, b_len, n_symbols)
    res = batch.graphs()
    batch.close()
    for i, (a, b) in enumerate(pairs):
        ref = O.lcs_graph(a, b)
        assert res.distance[i] == ref.distance and tuple(res.supremal[i]) == ref.supremal, i
        assert np.array_equal(res.nodes(i), ref.nodes) and np.array_equal(res.edges(i), ref.edges), i
    res.close()


@pytest.mark.parametrize("path", ["sweep", "bit-vector"])
def test_graphs_built_on_the_device_match_the_host_half(native, path, monkeypatch):
    """Kernel family 2c (malcs_gbuild.hip): ordering of the runs, dominance edges, splits, trimming and numbering on
    the device against the host half of the library (same pairs, MALCS_GBUILD=0) and against the oracle.  6 000 pairs:
    beyond 4 096 graphs of at most 32 runs the kernels run in four groups whose words come home on the second stream
    while the next group is built, followed by the group of the graphs with more runs.  Both device paths feed it: the
    sweeps their runs, the bit-vector kernels their events through k_events_to_runs."""
    from oracle import oracle as O
    data = gen.c2_batch(6000, seed=31)
    shifts = np.arange(6000, dtype=np.int64) % 7
    ctx = native.default_context()
    if path == "bit-vector":
        monkeypatch.setenv("MALCS_SWEEP_DMAX", "-1")
    batch = native.Batch(ctx, *data, 4)
    dev = batch.graphs(shifts)
    assert (dev.tie_pairs == 6000) == (path == "bit-vector")
    monkeypatch.setenv("MALCS_GBUILD", "0")
    host = batch.graphs(shifts)
    monkeypatch.delenv("MALCS_GBUILD")
    batch.close()
    try:
        assert dev.device_built == 6000 and host.device_built == 0
        assert np.array_equal(dev.counts, host.counts) and np.array_equal(dev.supremal, host.supremal)
        assert np.array_equal(dev.distance, host.distance)
        syms, a_off, a_len, b_off, b_len = data
        for i in range(6000):
            assert np.array_equal(dev.nodes(i), host.nodes(i)) and np.array_equal(dev.edges(i), host.edges(i)), i
            assert np.array_equal(dev.runs(i), host.runs(i)), i
            if i % 50 == 0:
                ref = O.lcs_graph(syms[int(a_off[i]):int(a_off[i]) + int(a_len[i])], syms[int(b_off[i]):int(b_off[i]) + int(b_len[i])], int(shifts[i]))
                assert tuple(dev.supremal[i]) == ref.supremal and np.array_equal(dev.nodes(i), ref.nodes) and np.array_equal(dev.edges(i), ref.edges)
                assert np.array_equal(dev.canonical(i), host.canonical(i))
    finally:
        dev.close()
        host.close()


@pytest.mark.parametrize("nsym", [4, 20])
def test_sweeps_with_several_pairs_per_wave(native, nsym, monkeypatch):
    """The one-wave sweeps take 8, 4 or 2 pairs per wave when the diagonals of every pair of a chunk fit a group's share
    of LDS (k_sweep_reach_multi / k_sweep_runs_multi); MALCS_SWEEP_PPW caps the number.  Same runs, same graphs for
    every cap -- on pairs from identical to unrelated, empty and one-symbol sequences, slides that outlast a lane's two
    words (identical 2.5 kb pairs), a pair count that leaves the last wave's groups empty -- and the oracle's graphs."""
    from oracle import oracle as O
    monkeypatch.setenv("MALCS_SWEEP_MIN_ROWS", "1")
    monkeypatch.setenv("MALCS_SWEEP_RATIO", "0")
    rs = np.random.RandomState(77 + nsym)
    syms, a_off, a_len, b_off, b_len = _window_class_batch(rs, 1203, nsym)
    a_len, b_len = a_len.copy(), b_len.copy()
    a_len[5], b_len[6], a_len[7], b_len[7] = 0, 0, 0, 0                      # empty on either side, on both
    a_len[9], b_len[10] = 1, 1
    ctx = native.default_context()
    results = {}
    for ppw in ("1", "2", "4", "8"):
        monkeypatch.setenv("MALCS_SWEEP_PPW", ppw)
        batch = native.Batch(ctx, syms, a_off, a_len, b_off, b_len, nsym)
        results[ppw] = batch.graphs()
        batch.close()
    try:
        one = results["1"]
        for ppw in ("2", "4", "8"):
            got = results[ppw]
            assert np.array_equal(got.distance, one.distance) and np.array_equal(got.counts, one.counts), ppw
            assert np.array_equal(got.supremal, one.supremal), ppw
            for i in range(1203):
                assert np.array_equal(got.runs(i), one.runs(i)), (ppw, i)
                assert np.array_equal(got.nodes(i), one.nodes(i)) and np.array_equal(got.edges(i), one.edges(i)), (ppw, i)
        got = results["8"]
        for i in list(range(0, 12)) + list(range(12, 1203, 40)):
            a = syms[int(a_off[i]):int(a_off[i]) + int(a_len[i])]
            b = syms[int(b_off[i]):int(b_off[i]) + int(b_len[i])]
            ref = O.lcs_graph(a, b, 0)
            assert int(got.distance[i]) == O.edit_distance(a, b), i
            assert tuple(got.supremal[i]) == ref.supremal and np.array_equal(got.nodes(i), ref.nodes) and np.array_equal(got.edges(i), ref.edges), i
    finally:
        for r in results.values():
            r.close()


def test_graphs_from_two_host_threads_with_a_context_each(native):
    """A context is a stream, a pool and worker threads of its own: two host threads, each with its context and batch,
    build their graphs side by side (ctypes releases the GIL for the call; one thread's host loops run under the
    other's kernels: 8.7 -> 6.6 ms per 100 000 C2 graphs, tools/time_graphs_in_flight.py) and get what one thread gets."""
    import threading
    datas = [gen.c2_batch(6000, seed=41 + t) for t in range(2)]
    ctxs = [native.Context(0) for _ in range(2)]
    alone = []
    for t in range(2):
        b = native.Batch(ctxs[t], *datas[t], 4)
        alone.append(b.graphs())
        b.close()
    together = [[], []]
    errors = []

    def work(t):
        try:
            for _ in range(3):
                b = native.Batch(ctxs[t], *datas[t], 4)
                together[t].append(b.graphs())
                b.close()
        except Exception as e:                                # noqa: BLE001 -- reported by the assert below
            errors.append(e)

    threads = [threading.Thread(target=work, args=(t,)) for t in range(2)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    try:
        assert not errors, errors
        for t in range(2):
            want = alone[t]
            for got in together[t]:
                assert got.device_built == 6000
                assert np.array_equal(got.distance, want.distance) and np.array_equal(got.counts, want.counts)
                assert np.array_equal(got.supremal, want.supremal)
                for i in range(0, 6000, 7):
                    assert np.array_equal(got.nodes(i), want.nodes(i)) and np.array_equal(got.edges(i), want.edges(i)), (t, i)
                    assert np.array_equal(got.runs(i), want.runs(i)), (t, i)
    finally:
        for g in alone + together[0] + together[1]:
            g.close()
        for c in ctxs:
            c.close()


def test_randomised_batches_through_every_switch(native):
    """tools/fuzz_parity.py's batches for fixed seeds: mixed shapes (1 ... 5 000 pairs, 0 ... 5 000 symbols, 2 ... 20 letters,
    identical to unrelated, tandem repeats), each through one of the nine settings of the switches that pick other kernels
    (seed mod 9), and a few batches of long pairs: every distance and a sample of graphs against the oracle."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("fuzz_parity", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "fuzz_parity.py"))
    fuzz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fuzz)
    ctx = native.default_context()
    graphs = 0
    for seed in range(100, 280):
        verdict, _, ng = fuzz.check_seed(ctx, seed)
        assert verdict is None, verdict
        graphs += ng
    for seed in range(300, 318):
        verdict, _, ng = fuzz.check_seed(ctx, seed, long=True)
        assert verdict is None, verdict
        graphs += ng
    assert graphs > 1500


@pytest.mark.parametrize("path", ["sweep", "band"])
def test_c5_ii_one_megabase_two_percent(native, path, monkeypatch):
    """BASELINE config 5 (ii): one 1 Mb pair at 2 % indels (D about 20 000) on both device paths.
    PARITY UNPINNED AGAINST THE REFERENCE: it cannot run this size (about 1e10 interpreted cell steps), and neither
    can the oracle's `_lcs_nodes` restatement, which keeps every run the sweep passes (killed at 64 GB here).
    What is checked instead: the distance against the oracle's O(NP) `edit_distance`; the graph against the
    oracle's independent `_build_graph` fed with the runs the device found; that those runs are matches, lie on
    consecutive levels and number LLCS levels; and the round trip -- the canonical allele patched into the reference
    spells the observed sequence, at a cost equal to the distance."""
    from oracle import oracle as O
    from algebra_amd.lcs.encoding import encode_pairs
    use_path(monkeypatch, path)
    rs = np.random.RandomState(55)
    a = gen.random_seq(rs, 1000000)
    b = gen.mutate(rs, a, 0.01, 0.01)
    ctx = native.default_context()
    batch = native.Batch(ctx, *encode_pairs([(a, b)]))
    res = batch.graphs()
    batch.close()
    try:
        want = O.edit_distance(a, b)
        assert int(res.distance[0]) == want and 15000 < want < 25000
        runs, nodes, edges = res.runs(0), res.nodes(0), res.edges(0)
        llcs = (len(a) + len(b) - want) // 2
        assert int(res.counts[0, 1]) == llcs
        # every run is a diagonal stretch of equal symbols ending on its level; together they cover all levels
        an, bn = np.frombuffer(a.encode(), np.uint8), np.frombuffer(b.encode(), np.uint8)
        covered = np.zeros(llcs + 1, np.int64)
        for row, col, length, level in runs[:: max(1, len(runs) // 4000)].tolist():
            assert np.array_equal(an[row:row + length], bn[col:col + length])
        np.add.at(covered, runs[:, 3] - runs[:, 2] + 1, 1)
        np.add.at(covered, runs[:, 3] + 1, -1)
        assert (np.cumsum(covered)[:llcs] > 0).all()
        ref = O.graph_from_runs(len(a), len(b), 0, runs, llcs, want)
        assert ref.supremal == tuple(res.supremal[0])
        assert np.array_equal(ref.nodes, nodes) and np.array_equal(ref.edges, edges)
        canon = res.canonical(0)
        pieces, pos, cost = [], 0, 0
        for start, end, lo, hi in canon.tolist():
            assert start >= pos
            pieces += [a[pos:start], b[lo:hi]]
            pos = end
            cost += (end - start) + (hi - lo)
        assert "".join(pieces + [a[pos:]]) == b and cost >= want
    finally:
        res.close()


def test_more_than_256_symbols(native):
    """Arbitrary str code points (reference edit_distance.py:26): a batch with hundreds of distinct symbols goes to the
    device with one code per symbol a pair's two sides share, and splits when even that is too many."""
    import algebra_amd
    from algebra_amd.lcs import LCSgraph
    from oracle import py_oracle as P
    rs = np.random.RandomState(41)
    pairs = []
    for k in range(40):
        alphabet = [chr(0x400 + 40 * k + i) for i in range(40)]          # 1 600 distinct symbols in the batch
        a = "".join(rs.choice(alphabet, 120))
        b = "".join(ch for ch in a if rs.random_sample() > 0.1) + "".join(rs.choice(alphabet, 5))
        pairs.append((a, b))
    want = [P.edit_distance(a, b) for a, b in pairs]
    assert algebra_amd.edit_distance_batch(pairs).tolist() == want
    graphs = LCSgraph.batch(pairs[:12])
    assert [g.distance for g in graphs] == want[:12]
    for (a, b), g in zip(pairs[:12], graphs):
        v = g.supremal
        assert a[:v.start] + v.sequence + a[v.end:] == b


def test_pairs_dealt_over_two_contexts(native):
    """The multi-device entry of the facade: pairs dealt over one context per listed device, results in
    the caller's order.  A test box has one GPU: the two contexts sit on device 0."""
    import algebra_amd
    from algebra_amd.lcs import LCSgraph
    from algebra_amd.lcs.encoding import deal
    from util import graph_record
    rs = np.random.RandomState(43)
    pairs = []
    for m in list(rs.randint(0, 3000, 300)) + [5000, 0, 64, 65]:
        a = gen.random_seq(rs, int(m))
        pairs.append((a, gen.mutate(rs, a, 0.02, 0.02)))
    parts = deal(pairs, 2)
    assert sorted(np.concatenate(parts).tolist()) == list(range(len(pairs))) and abs(len(parts[0]) - len(parts[1])) <= 1
    one = algebra_amd.edit_distance_batch(pairs)
    two = algebra_amd.edit_distance_batch(pairs, devices=[0, 0])
    assert np.array_equal(one, two)
    g1 = LCSgraph.batch(pairs[:60], shifts=range(60))
    g2 = LCSgraph.batch(pairs[:60], shifts=range(60), devices=[0, 0])
    assert [graph_record(g) for g in g1] == [graph_record(g) for g in g2]


def test_batch_create_rejects_codes_beyond_n_symbols(native):
    """The C-ABI does not trust `n_symbols`: a code the promised alphabet does not have is an error, not an alias of
    another symbol (n_symbols selects the bit planes; code 4 with two planes would read as code 0)."""
    ctx = native.default_context()
    syms = np.array([0, 1, 2, 3, 4, 1, 2, 3], np.uint8)
    args = (np.array([0], np.uint64), np.array([4], np.uint32), np.array([4], np.uint64), np.array([4], np.uint32))
    with pytest.raises(native.MalcsError, match="n_symbols"):
        native.Batch(ctx, syms, *args, 4)
    with pytest.raises(native.MalcsError, match="n_symbols"):
        native.Batch(ctx, np.array([0, 1, 2, 5, 6, 1, 2, 5], np.uint8), *args, 6)     # three planes hold 0 ... 7, the promise was < 6
    native.Batch(ctx, syms, *args, 5).close()
