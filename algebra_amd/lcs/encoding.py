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
        raise ValueError(f"{len(present)} distinct symbols in one batch; libmalcs packs at most {_MAX_SYMBOLS}")
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
