"""The entry point that allocates nothing (malcs_edit_distance_noalloc, include/malcs.h) against the existing
malcs_edit_distance_batch, through the C-ABI: two calls in a row on one GPU, the second with a smaller input
(the file-scope scratch array keeps the first call's contents), bit-exact.

The device buffers a caller of this entry point would already own come straight from the HIP runtime libmalcs itself
is linked against (hipMalloc / hipMemcpy through ctypes): no second framework initialises the GPU in this process."""

import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _pairs(rs, lengths, n_symbols, indel=0.02):
    seqs = []
    for m in lengths:
        a = rs.randint(0, n_symbols, m).astype(np.uint8)
        keep = rs.random_sample(m) > indel
        b = np.concatenate([a[keep], rs.randint(0, n_symbols, rs.randint(0, 4)).astype(np.uint8)])
        seqs += [a, b] if rs.randint(2) else [b, a]
    lens = np.array([len(x) for x in seqs], np.uint32)
    off = np.concatenate([[0], np.cumsum(lens[:-1], dtype=np.uint64)]).astype(np.uint64)
    syms = np.concatenate(seqs + [np.zeros(1, np.uint8)])
    return syms, off[0::2].copy(), lens[0::2].copy(), off[1::2].copy(), lens[1::2].copy()


def _existing(native, ctx, data, n_symbols):
    syms, a_off, a_len, b_off, b_len = data
    out = np.empty(len(a_off), np.int32)
    native.check(native.lib.malcs_edit_distance_batch(ctx.handle, native.ptr(syms), syms.size, native.ptr(a_off), native.ptr(a_len),
                                                      native.ptr(b_off), native.ptr(b_len), len(a_off), n_symbols, native.ptr(out)))
    return out


from util import hip_runtime as _runtime, to_device as _to_device  # noqa: E402


def _noalloc(native, ctx, data, n_symbols):
    hip = _runtime()
    p = [_to_device(np.ascontiguousarray(x)) for x in data]
    out = np.full(len(data[1]), -7, np.int32)
    d_out = _to_device(out)
    try:
        native.check(native.lib.malcs_edit_distance_noalloc(ctx.handle, None, p[0], data[0].size, p[1], p[2], p[3], p[4], len(data[1]),
                                                            n_symbols, d_out))
        ctx.sync()
        assert hip.hipMemcpy(out.ctypes.data_as(ctypes.c_void_p), d_out, out.nbytes, 2) == 0   # hipMemcpyDeviceToHost
    finally:
        for q in p + [d_out]:
            hip.hipFree(q)
    return out


def test_noalloc_matches_existing_entry_two_calls():
    from algebra_amd import _native as native
    ctx = native.default_context()
    rs = np.random.RandomState(77)
    # first call: every launch class (1 ... 16 384 symbols on the shorter side), DNA
    lengths = [0, 1, 31, 64, 65, 128, 129, 255, 256, 257, 300, 511, 513, 1000, 1024, 1025, 2047, 2049, 4096, 4097, 8191, 8193, 16384]
    lengths += list(rs.randint(0, 3000, 4000))
    big = _pairs(rs, lengths, 4)
    want = _existing(native, ctx, big, 4)
    got = _noalloc(native, ctx, big, 4)
    assert np.array_equal(got, want)
    # second call, smaller, other alphabet (three planes): the scratch array still holds the first call's lists
    small = _pairs(rs, list(rs.randint(0, 700, 300)) + [5000, 0, 33], 7, indel=0.1)
    want = _existing(native, ctx, small, 8)
    got = _noalloc(native, ctx, small, 8)
    assert np.array_equal(got, want)
    # and the first again: same answers as before
    assert np.array_equal(_noalloc(native, ctx, big, 4), _existing(native, ctx, big, 4))


def test_noalloc_limits():
    from algebra_amd import _native as native
    ctx = native.default_context()
    rs = np.random.RandomState(78)
    # beyond the maximum number of pairs: rejected with a clear error, nothing enqueued
    rc = native.lib.malcs_edit_distance_noalloc(ctx.handle, None, None, 0, None, None, None, None, 65537, 4, None)
    assert rc == -4 and b"at most 65536 pairs" in native.lib.malcs_last_error()
    rc = native.lib.malcs_edit_distance_noalloc(ctx.handle, None, None, 0, None, None, None, None, 1, 17, None)
    assert rc == -1
    # per pair on the device: a shorter side beyond 16 384 symbols, a sequence outside the buffer -> -1, the others exact
    data = list(_pairs(rs, [100, 17000, 200], 4))
    want = _existing(native, ctx, tuple(data), 4)
    data[3] = data[3].copy()
    data[3][2] = np.uint64(data[0].size)                       # pair 2: b starts at the end of the buffer and has symbols
    got = _noalloc(native, ctx, tuple(data), 4)
    assert got[0] == want[0] and got[1] == -1 and got[2] == -1


def test_noalloc_bad_codes_full_arena_and_four_planes():
    """A symbol code that is not below n_symbols marks the pairs of its sequence with -1 (malcs_batch_create rejects such
    a batch as a whole); sequences beyond the packed arena likewise, the pairs in front of them exact; 9 to 16 symbols
    (four planes) against the existing entry."""
    from algebra_amd import _native as native
    ctx = native.default_context()
    rs = np.random.RandomState(79)
    # four planes
    wide = _pairs(rs, list(rs.randint(0, 900, 500)) + [3000, 4097], 13, indel=0.05)
    assert np.array_equal(_noalloc(native, ctx, wide, 13), _existing(native, ctx, wide, 13))
    # codes out of range: beyond n_symbols but inside the planes (5 of 0..7 with n_symbols = 5), and beyond the planes
    data = list(_pairs(rs, [300, 400, 500, 600], 4))
    want = _existing(native, ctx, tuple(data), 4)
    syms = data[0].copy()
    syms[int(data[1][1]) + 7] = 6                               # pair 1, sequence a: 6 >= n_symbols = 5 (three planes hold it)
    syms[int(data[3][3]) + 100] = 9                             # pair 3, sequence b: beyond three planes
    data[0] = syms
    got = _noalloc(native, ctx, tuple(data), 5)
    assert got[0] == want[0] and got[2] == want[2] and got[1] == -1 and got[3] == -1
    # the packed arena (2 Mi dwords = 1 Mi blocks of 32 DNA symbols) is full after 2 x 16 pairs of 1 Mi symbols: the rest get -1
    n, length = 1200, 1 << 15
    a = rs.randint(0, 4, length).astype(np.uint8)
    syms = np.concatenate([a, a[: length - 5]])
    a_off = np.zeros(n, np.uint64); a_len = np.full(n, length, np.uint32)
    b_off = np.full(n, length, np.uint64); b_len = np.full(n, length - 5, np.uint32)
    a_len[0] = b_len[0] = 1000                                  # (a pair that any arena holds)
    got = _noalloc(native, ctx, (syms, a_off, a_len, b_off, b_len), 4)
    assert got[0] == 0                                           # prefixes of the same sequence
    assert (got[1:] == -1).all()                                 # shorter side beyond 16 384 symbols: limit per pair (32 768 here)
    short = 16000
    a_len[:] = short; b_len[:] = short - 5
    n_fit = (2 << 20) // 2 // (2 * ((short + 31) // 32))         # pairs whose blocks the arena holds
    got = _noalloc(native, ctx, (syms, a_off, a_len, b_off, b_len), 4)
    assert (got >= 0).all() if n <= n_fit else ((got[:n_fit] == 5).all() and (got[n_fit + 1:] == -1).all())
