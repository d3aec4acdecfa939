"""Dev tool: randomised differential run of the library against the oracle (distances of every pair, graphs of a sample),
over batches of mixed shapes and the environment switches that pick other kernels.  On the GPU box:
    python tools/fuzz_parity.py [SECONDS] [FIRST_SEED] [long]
Prints one line per batch; stops at the first difference (exit code 1) with the seed, the switches and the pair."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from algebra_amd import _native as N
from oracle import oracle as O

SWITCHES = [{}, {"MALCS_SWEEP_MIN_ROWS": "1", "MALCS_SWEEP_RATIO": "0"}, {"MALCS_SWEEP_MIN_ROWS": "1", "MALCS_SWEEP_RATIO": "0", "MALCS_SWEEP_PPW": "2"},
            {"MALCS_SWEEP_DMAX": "-1"}, {"MALCS_DIAG_HALF": "0"}, {"MALCS_DIAG_NLW": "0"}, {"MALCS_SAMPLE_WAVE": "1"}, {"MALCS_GBUILD": "0"},
            {"MALCS_SWEEP_MIN_ROWS": "1", "MALCS_SWEEP_RATIO": "0", "MALCS_SWEEP_PPW": "4"}]


def batch_of(rs, long=False):
    """long: few pairs of up to 70 000 symbols (big sweeps, band kernels, sliding windows)"""
    nsym = int(rs.choice([2, 3, 4, 4, 4, 5, 13, 20]))
    n = int(rs.choice([1, 7, 63, 64, 65, 300, 1500, 5000]))
    top = int(rs.choice([8, 70, 200, 700, 1100, 2600, 5000]))
    if long:
        nsym, n, top = 4, int(rs.choice([1, 3, 8])), int(rs.choice([5000, 20000, 70000]))
    seqs = []
    for i in range(n):
        length = int(rs.randint(0, top + 1))
        a = rs.randint(0, nsym, length).astype(np.uint8)
        kind = rs.randint(0, 10)
        if kind == 0:
            b = a.copy()
        elif kind == 1:
            b = rs.randint(0, nsym, int(rs.randint(0, top + 1))).astype(np.uint8)
        elif kind == 2 and length > 20:                       # a tandem repeat whose copy number changes
            unit = a[:int(rs.randint(1, 6))]
            at = int(rs.randint(0, length))
            a = np.concatenate([a[:at], np.tile(unit, int(rs.randint(2, 12))), a[at:]])
            b = np.concatenate([a[:at], np.tile(unit, int(rs.randint(0, 14))), a[at + len(unit) * 3:]])
        else:
            p = float(rs.choice([0.0002, 0.001, 0.004] if long else [0.001, 0.005, 0.02, 0.05, 0.15]))
            keep = rs.random_sample(length) >= p
            ins = rs.random_sample(length) < p
            where = np.repeat(np.arange(length), ins.astype(np.int64) + keep.astype(np.int64))      # an inserted symbol, then the kept one
            fresh = np.ones(len(where), bool)
            fresh[np.cumsum(ins.astype(np.int64) + keep.astype(np.int64))[keep] - 1] = False
            b = np.where(fresh, rs.randint(0, nsym, len(where)), a[where] if length else 0).astype(np.uint8)
        if rs.randint(0, 2):
            a, b = b, a
        seqs.append((a, b))
    syms = np.concatenate([x for ab in seqs for x in ab] + [np.zeros(1, np.uint8)]).astype(np.uint8)
    a_off, a_len, b_off, b_len = [], [], [], []
    pos = 0
    for a, b in seqs:
        a_off.append(pos); a_len.append(len(a)); pos += len(a)
        b_off.append(pos); b_len.append(len(b)); pos += len(b)
    return (syms, np.array(a_off, np.uint64), np.array(a_len, np.uint32), np.array(b_off, np.uint64), np.array(b_len, np.uint32)), nsym


def check_seed(ctx, seed, long=False):
    """One batch: None if the library and the oracle agree, else what differs.  -> (verdict, pairs, graphs compared)"""
    rs = np.random.RandomState(seed)
    data, nsym = batch_of(rs, long)
    env = SWITCHES[seed % len(SWITCHES)]
    for k, v in env.items():
        os.environ[k] = v
    graphs = 0
    try:
        syms, a_off, a_len, b_off, b_len = data
        n = len(a_len)
        want = O.edit_distance_batch(*data)
        b = N.Batch(ctx, *data, nsym)
        try:
            got = b.distances()
            if not np.array_equal(got, want):
                i = int(np.nonzero(got != want)[0][0])
                return "DISTANCE DIFFERS seed %d %s pair %d got %d want %d lengths %d %d nsym %d" % (seed, env, i, got[i], want[i], a_len[i], b_len[i], nsym), n, graphs
            shifts = rs.randint(0, 5, n).astype(np.int64)
            # graphs up to 300 edits only: the graph of two unrelated sequences over two symbols has 7e7 edges (the
            # reference's too), a batch of them more than the box's memory
            g = b.graphs(shifts, np.full(n, 300, np.int64))
            try:
                if not np.array_equal(g.distance, want) or not np.array_equal(g.status != 0, want > 300):
                    return "GRAPH DISTANCE OR STATUS DIFFERS seed %d %s" % (seed, env), n, graphs
                for i in rs.choice(n, min(n, 40), replace=False):
                    i = int(i)
                    if want[i] > 300:
                        continue
                    a = syms[int(a_off[i]):int(a_off[i]) + int(a_len[i])]; c = syms[int(b_off[i]):int(b_off[i]) + int(b_len[i])]
                    ref = O.lcs_graph(a, c, int(shifts[i]))
                    if tuple(g.supremal[i]) != ref.supremal or not np.array_equal(g.nodes(i), ref.nodes) or not np.array_equal(g.edges(i), ref.edges):
                        return "GRAPH DIFFERS seed %d %s pair %d lengths %d %d distance %d nsym %d" % (seed, env, i, a_len[i], b_len[i], want[i], nsym), n, graphs
                    graphs += 1
            finally:
                g.close()
        finally:
            b.close()
        return None, n, graphs
    finally:
        for k in env:
            del os.environ[k]


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    long = len(sys.argv) > 3 and sys.argv[3] == "long"
    ctx = N.Context(0)
    t0 = time.time()
    batches = pairs = graphs = 0
    while time.time() - t0 < budget:
        verdict, n, ng = check_seed(ctx, seed, long)
        if verdict:
            print(verdict, flush=True)
            return 1
        batches += 1; pairs += n; graphs += ng
        print("seed", seed, "pairs", n, SWITCHES[seed % len(SWITCHES)], "ok", "%.0f s" % (time.time() - t0), flush=True)
        seed += 1
    print("done:", batches, "batches,", pairs, "distances,", graphs, "graphs against the oracle, no difference", flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
