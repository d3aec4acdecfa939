"""Simple (insert/delete-only) edit distance on the GPU.

Drop-in for the reference's `algebra/lcs/edit_distance.py:21-53`; the batched form is what the
relation drivers use.  Both run kernel family 1 of libmalcs (no CPU path).
"""

import numpy as np

from .. import _native
from .encoding import TooManySymbols, deal, encode_pairs


def edit_distance_batch(pairs, ctx=None, devices=None):
    """Distances of many (a, b) pairs -> numpy int32 array, in the order of `pairs`.

    One device batch on `ctx` (default: this process's context), or, with `devices` = [device ids] (default:
    MALCS_DEVICES when set), the pairs dealt over one context per entry -- same mix of lengths for each, no exchange
    between them -- and every device's results written into its own places of the one output array."""
    pairs = list(pairs)
    contexts = _native.contexts_for(devices) if ctx is None else None
    if contexts and len(contexts) > 1 and len(pairs) > 1:
        out = np.empty(len(pairs), dtype=np.int32)

        def part(job):
            context, index = job
            if len(index):
                out[index] = edit_distance_batch([pairs[i] for i in index.tolist()], ctx=context)
        _native.run_on_devices(part, list(zip(contexts, deal(pairs, len(contexts)))))
        return out
    ctx = ctx or (contexts[0] if contexts else _native.default_context())
    try:
        syms, a_off, a_len, b_off, b_len, n_symbols = encode_pairs(pairs)
    except TooManySymbols:
        if len(pairs) < 2:
            raise
        half = len(pairs) // 2                  # fewer pairs share fewer symbols
        return np.concatenate([edit_distance_batch(pairs[:half], ctx=ctx), edit_distance_batch(pairs[half:], ctx=ctx)])
    batch = _native.Batch(ctx, syms, a_off, a_len, b_off, b_len, n_symbols)
    try:
        return batch.distances()
    finally:
        batch.close()


def edit_distance(a, b):
    """Calculate the simple edit distance between two strings."""
    return int(edit_distance_batch([(a, b)])[0])
