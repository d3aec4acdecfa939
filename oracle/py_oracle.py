"""Pure-Python restatement of the reference's edit_distance -- TEST INFRASTRUCTURE ONLY.

Follows algebra/lcs/edit_distance.py:21-53 (Wu-Manber-Myers O(NP)) on str or integer arrays.  It
exists so that bench.py can time "the reference's CPU path" (interpreted Python, one process per
core) on the GPU box, where the reference itself cannot travel; tests/test_oracle.py pins it to
the golden vectors.  Nothing under algebra_amd/ imports it.
"""


def edit_distance(a, b):
    m, n = len(a), len(b)
    if m >= n:                                   # :33-35
        a, b, m, n = b, a, n, m
    offset, delta = m + 1, n - m                 # :37-38
    fp = [-1] * (m + n + 3)                      # :39

    def snake(k):                                # :23-29
        col = max(fp[k - 1 + offset] + 1, fp[k + 1 + offset])
        row = col - k
        while row < m and col < n and a[row] == b[col]:
            row += 1
            col += 1
        fp[k + offset] = col

    p = -1
    while True:                                  # :42-51
        p += 1
        for k in range(-p, delta):
            snake(k)
        for k in range(delta + p, delta, -1):
            snake(k)
        snake(delta)
        if fp[delta + offset] >= n:
            return delta + 2 * p                 # :53


def edit_distance_slice(args):
    """Worker for multiprocessing: distances of pairs [lo, hi) of a flat batch."""
    syms, a_off, a_len, b_off, b_len, lo, hi = args
    raw = bytes(syms)
    out = []
    for i in range(lo, hi):
        a = raw[a_off[i]:a_off[i] + a_len[i]]
        b = raw[b_off[i]:b_off[i] + b_len[i]]
        out.append(edit_distance(a, b))
    return out
