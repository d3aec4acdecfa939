"""Simple (insert/delete-only) edit distance on the GPU.

Drop-in for the reference's `algebra/lcs/edit_distance.py:21-53`; the batched form is what the
relation drivers use.  Both run kernel family 1 of libmalcs (no CPU path).
"""

from .. import _native
from .encoding import encode_pairs


def edit_distance_batch(pairs, ctx=None):
    """Distances of many (a, b) pairs in one device batch -> numpy int32 array."""
    pairs = list(pairs)
    ctx = ctx or _native.default_context()
    syms, a_off, a_len, b_off, b_len, n_symbols = encode_pairs(pairs)
    batch = _native.Batch(ctx, syms, a_off, a_len, b_off, b_len, n_symbols)
    try:
        return batch.distances()
    finally:
        batch.close()


def edit_distance(a, b):
    """Calculate the simple edit distance between two strings."""
    return int(edit_distance_batch([(a, b)])[0])
