"""Symbols -> uint8 codes for the device.

The reference compares Python str code points with == (algebra/lcs/edit_distance.py:26), so any
injective map of the symbols that occur in a batch preserves every result.  A, C, G, T get codes
0..3 so that plain DNA needs two bit planes on the device.
"""

import numpy as np

_MAX_SYMBOLS = 256


def _raw(seq):
    """str -> array of code points (uint8 when Latin-1, else uint32)."""
    if isinstance(seq, (bytes, bytearray)):
        return np.frombuffer(bytes(seq), dtype=np.uint8)
    try:
        return np.frombuffer(seq.encode("latin-1"), dtype=np.uint8)
    except UnicodeEncodeError:
        return np.frombuffer(seq.encode("utf-32-le"), dtype=np.uint32)


class TooManySymbols(ValueError):
    """More distinct symbols than the device's 8 bit planes hold, even after reducing every pair to the symbols its
    two sequences share.  Batch callers split the batch and try again; for a single pair it is final."""


def _encode_shared_only(pairs):
    """The batch has more than 256 distinct symbols.  Only equality between a symbol of `a` and a symbol of `b`
    of the same pair matters (reference edit_distance.py:26, lcs_graph.py:287), so a symbol that occurs on one side
    only of every pair it is in needs no code of its own: all such symbols of an `a` get one code, those of a `b`
    another.  Sequences are not shared between pairs here (one stored sequence could otherwise serve as `a` and as
    `b`)."""
    pairs = [tuple(pair) for pair in pairs]
    shared = set()
    for a, b in pairs:
        shared |= set(a) & set(b)
    if len(shared) + 2 > _MAX_SYMBOLS:
        raise TooManySymbols(f"{len(shared)} distinct symbols are shared inside the pairs of one batch; libmalcs packs at most "
                             f"{_MAX_SYMBOLS - 2} of them")
    table = {sym: code for code, sym in enumerate(sorted(shared, key=repr))}
    only_a, only_b = len(table), len(table) + 1
    chunks, lens = [], []
    for a, b in pairs:
        chunks.append(np.fromiter((table.get(x, only_a) for x in a), dtype=np.uint8, count=len(a)))
        chunks.append(np.fromiter((table.get(x, only_b) for x in b), dtype=np.uint8, count=len(b)))
        lens += [len(a), len(b)]
    lens = np.array(lens, dtype=np.uint64).reshape(-1, 2)
    offs = np.concatenate([[0], np.cumsum(lens.reshape(-1))[:-1]]).astype(np.uint64).reshape(-1, 2)
    syms = np.concatenate(chunks + [np.zeros(1, np.uint8)])
    return (syms, np.ascontiguousarray(offs[:, 0]), np.ascontiguousarray(lens[:, 0]).astype(np.uint32),
            np.ascontiguousarray(offs[:, 1]), np.ascontiguousarray(lens[:, 1]).astype(np.uint32), len(table) + 2)


def deal(pairs, n_parts):
    """Indices of `pairs` for each of `n_parts` devices: pairs sorted by the 64-row word count of their
    shorter sequence, dealt round-robin, so that every device gets the same mix of launch classes."""
    words = np.fromiter(((min(len(a), len(b)) + 63) // 64 for a, b in pairs), dtype=np.int64, count=len(pairs))
    order = np.argsort(words, kind="stable")
    return [order[k::n_parts] for k in range(n_parts)]


def encode_pairs(pairs):
    """[(reference, observed), ...] -> (syms, a_off, a_len, b_off, b_len, n_symbols).

    Sequences are interned by identity/value so that a sequence shared by many pairs is stored once.
    """
    index = {}
    unique = []
    refs = []
    for pair in pairs:
        for seq in pair:
            slot = index.get(seq)
            if slot is None:
                slot = index[seq] = len(unique)
                unique.append(seq)
            refs.append(slot)
    lens = np.fromiter((len(x) for x in unique), dtype=np.uint64, count=len(unique))
    offs = np.zeros(len(unique) + 1, dtype=np.uint64)
    np.cumsum(lens, out=offs[1:])
    flat = None
    if unique and all(isinstance(x, str) for x in unique):
        try:                                   # one encode for the whole batch (Latin-1 covers every alphabet seen so far)
            flat = np.frombuffer("".join(unique).encode("latin-1"), dtype=np.uint8)
        except UnicodeEncodeError:
            flat = None
    if flat is None:
        arrays = [_raw(x) for x in unique]
        if arrays and any(x.dtype != np.uint8 for x in arrays):
            flat = np.concatenate([x.astype(np.uint32) for x in arrays])
        else:
            flat = np.concatenate(arrays) if arrays else np.zeros(0, np.uint8)
    if flat.dtype != np.uint8:
        present = np.unique(flat)
    else:
        present = np.nonzero(np.bincount(flat, minlength=256))[0]
    dna = [ord(ch) for ch in "ACGT"]
    others = [int(v) for v in present if int(v) not in dna]
    if len(others) + 4 > _MAX_SYMBOLS and len(present) > _MAX_SYMBOLS:
        return _encode_shared_only(pairs)
    table = {v: i for i, v in enumerate(dna)}
    for v in others:
        table[v] = len(table)
    n_symbols = 4 if not others else len(table)
    if n_symbols > _MAX_SYMBOLS:
        # more than 252 non-DNA symbols: drop the reserved DNA codes that are unused
        table = {int(v): i for i, v in enumerate(present)}
        n_symbols = len(table)
    if flat.dtype == np.uint8:
        lut = np.zeros(256, dtype=np.uint8)
        for v, code in table.items():
            if v < 256:
                lut[v] = code
        syms = lut[flat]
    else:
        keys = np.array(sorted(table), dtype=np.uint32)
        vals = np.array([table[int(k)] for k in keys], dtype=np.uint8)
        syms = vals[np.searchsorted(keys, flat)]
    refs = np.array(refs, dtype=np.int64).reshape(-1, 2) if refs else np.zeros((0, 2), dtype=np.int64)
    a_off, b_off = offs[refs[:, 0]], offs[refs[:, 1]]
    a_len, b_len = lens[refs[:, 0]].astype(np.uint32), lens[refs[:, 1]].astype(np.uint32)
    if len(syms) == 0:
        syms = np.zeros(1, dtype=np.uint8)
    return syms, a_off, a_len, b_off, b_len, n_symbols
