#!/usr/bin/env python3
"""bench.py -- BASELINE config 2: batched LCS edit distance over 1 kb windows on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--pairs P]

One "step" is what a caller pays who computes the distances of a batch ONCE: the whole plan of the batch (on the device:
common ends of every pair, sort into tiles, mode of the class from a sample, pieces and their deal to the SIMDs, rows of
every piece) + one pass of kernel family 1, over a batch of P = 100 000 pairs of ~1 kb windows with ~2 % indels (SURVEY
8(d) C2).  Resident in HBM when the timed region starts: the packed bit-plane arena (2 bit per base) and the per-pair
offset / length tables -- the inputs, nothing derived from what the sequences hold.  With N > 1 there is
one rank per GPU: started by a launcher (torch.distributed.run sets RANK / WORLD_SIZE / MASTER_*) or, when there is
none, by this script itself (N child processes, started before anything touches the GPU); every rank owns its own batch of P pairs
(independent pairs, no data-path collective => weak scaling) and the job rate is all pairs of
all ranks over the slowest rank's time.

Prints ONE JSON line (rank 0).  `value` is whole-job GCUPS = sum(m_i * n_i) / seconds / 1e9
(unpadded full-matrix-equivalent cells); pairs/s is printed beside it.
`roofline` is for the dominant kernel (the window kernel k_llcs_diag when the planner puts the batch
into the window class, else k_llcs_bitpar), timed with HIP events on the library's stream; `cpu_baseline` times the CPU oracle (port of the reference's O(NP) algorithm) on the host.
"""

import argparse
import json
import multiprocessing
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

PEAK_LANE_OPS = 256 * 4 * 32 * 2.4e9          # 256 CU x 4 SIMD x 32 lanes/clk x 2.4 GHz = 78.6 T int32 lane-ops/s
OPS_PER_CELL = 0.125                           # 4 dword ops per 32 cells


_DATA = None        # the batch, inherited by the forked CPU-baseline workers (no pickling of 200 MB)


def _oracle_pass(bounds):
    from oracle import oracle as O
    syms, a_off, a_len, b_off, b_len = _DATA
    lo, hi = bounds
    O.edit_distance_batch(syms, a_off[lo:hi], a_len[lo:hi], b_off[lo:hi], b_len[lo:hi])
    return hi - lo


def _python_pass(bounds):
    from oracle import py_oracle as P
    syms, a_off, a_len, b_off, b_len = _DATA
    lo, hi = bounds
    return len(P.edit_distance_slice((syms[:int(b_off[hi - 1]) + int(b_len[hi - 1])], a_off.tolist(), a_len.tolist(),
                                      b_off.tolist(), b_len.tolist(), lo, hi)))


def cpu_baseline(data, cells_per_pair, budget_s=12.0):
    """Oracle (C port of the reference's O(NP) edit_distance) on the host cores, bounded sample."""
    global _DATA
    _DATA = data
    cores = max(1, min(16, os.cpu_count() or 1))
    n = len(data[1])
    bounds = np.linspace(0, n, cores + 1).astype(int)
    jobs = [(int(bounds[i]), int(bounds[i + 1])) for i in range(cores)]
    ctx = multiprocessing.get_context("fork")
    with ctx.Pool(cores) as pool:
        pool.map(_oracle_pass, jobs)                       # warm-up (library load, page faults)
        passes, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < budget_s and passes < 1000:
            pool.map(_oracle_pass, jobs)
            passes += 1
        wall = time.perf_counter() - t0
        # the reference's own flavour: interpreted Python, one process per core, smaller sample
        py_n = min(n, 4096 * cores)
        pb = np.linspace(0, py_n, cores + 1).astype(int)
        t1 = time.perf_counter()
        pool.map(_python_pass, [(int(pb[i]), int(pb[i + 1])) for i in range(cores)])
        py_wall = time.perf_counter() - t1
    pairs = passes * n
    return {
        "value": round(pairs * cells_per_pair / wall / 1e9, 3), "unit": "GCUPS", "cores": cores, "kind": "port",
        "pairs_per_s": round(pairs / wall, 1),
        "sample": f"{passes} passes over all {n} pairs of the workload, oracle/lcs_oracle.c O(NP) edit_distance, "
                  f"{cores} processes, {wall:.1f} s wall",
        "python_restatement": {"value": round(py_n * cells_per_pair / py_wall / 1e9, 3), "unit": "GCUPS",
                               "pairs_per_s": round(py_n / py_wall, 1), "cores": cores,
                               "sample": f"{py_n} pairs, oracle/py_oracle.py (interpreted, as the reference), {py_wall:.1f} s wall"},
    }


def extra_configs(ctx, _native, gen):
    """Secondary figures beside the headline (not the timed region): kernel family 2 on the C2 batch shape
    and the C4 relation matrix, both end to end from host buffers."""
    out = {}
    # the headline batch over the full matrix (window class switched off): the kernel every pair falls
    # back to when it is not similar enough for the window
    os.environ["MALCS_DIAG_NLW"] = "0"
    full = _native.Batch(ctx, *gen.c2_batch(100000, seed=2), 4)
    del os.environ["MALCS_DIAG_NLW"]
    full.time_distances(5)
    ms = full.time_distances(20) / 20
    fc = float(full.plan_info()["full_cells"])
    out["c2_full_matrix_kernel"] = {"kernel": "k_llcs_bitpar<4,2,true>", "kernel_ms": round(ms, 4),
                                    "GCUPS": round(fc / (ms * 1e-3) / 1e9, 1),
                                    "roofline_frac": round(fc * OPS_PER_CELL / (ms * 1e-3) / PEAK_LANE_OPS, 4)}
    full.close()
    # ... and through the window class with three limbs for every tile (what a less similar batch gets: T = 64)
    os.environ["MALCS_DIAG_HALF"] = "0"
    three = _native.Batch(ctx, *gen.c2_batch(100000, seed=2), 4)
    del os.environ["MALCS_DIAG_HALF"]
    three.time_dominant_kernel(20)
    ms = three.time_dominant_kernel(200) / 200
    wc = float(three.plan_info()["window_cells"])
    out["c2_three_limb_window"] = {"kernel": "k_llcs_diag<3,2,true>, no tile in two-limb mode", "kernel_ms": round(ms, 4),
                                   "window_cells": wc, "roofline_frac": round(wc * OPS_PER_CELL / (ms * 1e-3) / PEAK_LANE_OPS, 4)}
    three.close()
    n = 100000
    batch = _native.Batch(ctx, *gen.c2_batch(n, seed=12), 4)
    for _ in range(4):
        batch.graphs().close()                                 # warm-up (code objects; the pools of pinned and device blocks fill; the allocator settles after three calls)
    walls = []
    for _ in range(5):
        t0 = time.perf_counter()
        g = batch.graphs()
        walls.append(time.perf_counter() - t0)
        if len(walls) < 5:
            g.close()
    wall = sorted(walls)[2]
    out["c2_lcs_graphs"] = {"pairs": n, "graphs_per_s": round(n / wall, 1), "wall_ms_of_five_calls": [round(w * 1e3, 2) for w in walls],
                            "device_path": "one-wave sweeps, up to 8 pairs per wave (k_sweep_reach_multi + k_sweep_runs_multi)",
                            "backward_sweep_ms": round(float(g.timing[0]), 3), "forward_sweep_ms": round(float(g.timing[1]), 3),
                            "graphs_built_on_device": int(g.device_built), "device_build_ms": round(float(g.device_build_ms), 3),
                            "host_build_ms": round(float(g.timing[2]), 1)}
    g.close()
    batch.close()
    # the same shape through the bit-vector kernels (what short or divergent windows use).  Since round 3 windows that
    # are close to their reference store the tie bits of the band only (groups of 64 rows; the distance is known from
    # family 1): 25.7 -> 3.2 GB per 100 000 pairs, and the two kernels are no longer bound by HBM.  The second run stores
    # every group (MALCS_GRAPH_RESIDENT_BAND=0): that one IS the HBM-bound case, with family 2's own roofline
    # (0.25 B of tie bits per cell written by the forward pass, read by the backward pass; profiles/r03_family2_*).
    n = 100000
    cells = float(n) * 1e6
    for key, env in (("c2_lcs_graphs_bitvector", None), ("c2_lcs_graphs_bitvector_all_groups", "0")):
        os.environ["MALCS_SWEEP_DMAX"] = "-1"
        if env is not None:
            os.environ["MALCS_GRAPH_RESIDENT_BAND"] = env
        try:
            batch = _native.Batch(ctx, *gen.c2_batch(n, seed=12), 4)
            batch.graphs().close()
            t0 = time.perf_counter()
            g = batch.graphs()
            wall = time.perf_counter() - t0
        finally:
            del os.environ["MALCS_SWEEP_DMAX"]
            os.environ.pop("MALCS_GRAPH_RESIDENT_BAND", None)
        fwd, rch, tie = float(g.timing[0]), float(g.timing[1]), float(g.tie_bytes)
        rec = {"pairs": n, "graphs_per_s": round(n / wall, 1), "forward_ms": round(fwd, 3), "reach_ms": round(rch, 3),
               "host_build_ms": round(float(g.timing[2]), 1), "tie_bytes": int(tie),
               "cells_per_s": {"forward": round(cells / (fwd * 1e-3), 1), "reach": round(cells / (rch * 1e-3), 1)},
               "tie_bit_GB_per_s": {"forward_write": round(tie / (fwd * 1e-3) / 1e9, 1), "reach_read": round(tie / (rch * 1e-3) / 1e9, 1)}}
        for stage, ms, kernel in (("forward", fwd, "k_lcs_forward<W,2,false>"), ("reach", rch, "k_lcs_reach<W,2,false>")):
            gbs = tie / (ms * 1e-3) / 1e9
            rec["roofline_" + stage] = {"bound": "hbm" if env == "0" else "valu (the band's tie bits no longer fill the link; the HBM figure is kept beside it)",
                                        "kernel": kernel, "achieved": round(gbs, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(gbs / 8000.0, 4)}
        out[key] = rec
        g.close()
        batch.close()
    return out


def long_window_configs(ctx, _native, gen):
    """BASELINE configs 3 and 5: LCS graphs of long windows.  Two device paths build them (identical
    graphs): the output-sensitive sweeps (default for similar windows: no predecessor masks at all, a table of furthest
    points of (D + 1)(D + 2) / 2 words per pair written by the backward sweep and read by the forward one) and the
    bit-vector kernels (2 bits of tie masks per band cell, written by the forward pass, read by the backward pass)."""
    from algebra_amd.lcs.encoding import encode_pairs
    out = {}

    def run(pairs, reps, bitvector):
        if bitvector:
            os.environ["MALCS_SWEEP_DMAX"] = "-1"
        try:
            syms, ao, al, bo, bl, ns = encode_pairs(pairs)
            cells = float((al.astype(np.float64) * bl).sum())
            bt = _native.Batch(ctx, syms, ao, al, bo, bl, ns)
            bt.graphs().close()                                    # warm-up
            walls = []
            for _ in range(reps):
                t0 = time.perf_counter()
                g = bt.graphs()
                walls.append(time.perf_counter() - t0)
                if len(walls) < reps:
                    g.close()
            wall = sorted(walls)[len(walls) // 2]
            dist = g.distance.astype(np.float64)
            rec = {"pairs": len(pairs), "cells_full_matrix": cells, "cells_per_s": round(cells / wall, 1), "wall_ms": round(wall * 1e3, 2),
                   "distance": [int(x) for x in g.distance[:4]], "nodes": [int(x) for x in g.counts[:4, 2]], "edges": [int(x) for x in g.counts[:4, 3]],
                   "pass1_ms": round(float(g.timing[0]), 3), "pass2_ms": round(float(g.timing[1]), 3),
                   "build_device_ms": round(float(g.device_build_ms), 3), "graphs_built_on_device": int(g.device_built),
                   "build_host_ms": round(float(g.timing[2]), 2)}
            rec["other_ms"] = round(rec["wall_ms"] - rec["pass1_ms"] - rec["pass2_ms"] - rec["build_host_ms"], 2)   # copies, allocation, read-out
            if bitvector:
                rec["device_path"] = "bit-vector kernels k_lcs_forward / k_lcs_reach (band from the exact distance)"
                rec["tie_mask_bytes_written"] = int(g.tie_bytes)
                rec["tie_mask_bytes_read"] = int(g.tie_bytes)
                rec["band_bytes_algorithmic"] = "0.25 B per band cell: the rows are the band"
                for stage, key in (("forward_write", "pass1_ms"), ("reach_read", "pass2_ms")):
                    if rec[key] > 0:
                        gbs = g.tie_bytes / (rec[key] * 1e-3) / 1e9
                        rec["roofline_" + stage] = {"bound": "hbm", "achieved": round(gbs, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(gbs / 8000.0, 4)}
            else:
                table = float(((dist + 1) * (dist + 2) / 2 * 4).sum())
                rec["device_path"] = "furthest-reaching sweeps k_sweep_reach (backwards, keeps the table) / k_sweep_runs (forwards); no predecessor masks"
                rec["tie_mask_bytes_written"] = 0
                rec["reach_table_bytes"] = int(table)
                rec["bound"] = "latency of one workgroup per pair (D + 1 dependent budgets); not a bandwidth kernel"
            g.close()
            bt.close()
            return rec
        finally:
            os.environ.pop("MALCS_SWEEP_DMAX", None)

    c3 = [gen.tandem_pair(np.random.RandomState(300 + s), 10000, 20) for s in range(16)]
    out["c3"] = {"workload": "16 pairs, 10 kb reference with 20 tandem-repeat tracts, copy numbers changed + 0.5 % indels (tests/gen.py tandem_pair, seeds 300-315)",
                 "sweeps": run(c3, 5, False), "bit_vector": run(c3, 3, True)}
    sparse = [gen.large_pair({"kind": "few", "seed": 50 + i, "length": 1000000, "events": 20}) for i in range(4)]
    rs = np.random.RandomState(5)
    a = gen.random_seq(rs, 1000000)
    dense = [(a, gen.mutate(rs, a, 0.01, 0.01))]
    out["c5"] = {"workload": "1 Mb pairs: (i) 4 pairs with 20 small indels, (ii) 1 pair at 2 % indels",
                 "i_sweeps": run(sparse, 3, False), "i_bit_vector": run(sparse, 2, True),
                 "ii_sweeps": run(dense, 2, False), "ii_bit_vector": run(dense, 1, True)}
    return out


def relation_matrix(ctx, gen, rank, world, comm, gather, fence, all_max):
    """BASELINE config 4, the other half of the metric: the 5 000 x 5 000 `graph_based.compare` matrix over clustered
    variants in a 100 kb locus.  Upper-triangular 256 x 256 tiles are dealt round-robin to the ranks; the one
    collective is the all-gather of the uint8 label tiles (ncclAllGather, called by the library on its own buffers:
    `comm`; the gloo rehearsal on one card passes `gather` instead); every rank ends with the full matrix.  Strong
    scaling of a 12 ms job: the per-rank share shrinks, preparation / all-gather / assembly do not."""
    from algebra_amd.lcs import LCSgraph
    from algebra_amd.relations import compare_matrix
    from algebra_amd.relations.matrix import columns_of
    from algebra_amd.variants import Variant
    case = gen.matrix_case(seed=4, length=100000, n_clusters=50, per_cluster=100, width=200)
    variants = [[Variant(*v)] for v in case["variants"]]
    LCSgraph.from_variant_batch(case["reference"], variants, ctx=ctx)     # warm-up (code objects; the pools of pinned and device blocks fill)
    t_calls = []
    for _ in range(3):
        t0 = time.perf_counter()
        graphs = LCSgraph.from_variant_batch(case["reference"], variants, ctx=ctx)
        t_calls.append(time.perf_counter() - t0)
    t_graphs = sorted(t_calls)[1]
    # the per-variant columns of the matrix (intervals, distances, supremal keys, symbol and atom masks) belong to the
    # batch of graphs, which keeps them: made here, once, and reported beside the graphs -- not part of `matrix_s`
    t0 = time.perf_counter()
    graphs.columns(case["reference"], columns_of)
    t_columns = time.perf_counter() - t0
    how = {"comm": comm, "rank": rank, "world": world, "gather": gather}
    compare_matrix(case["reference"], graphs, **how)           # warm-up (code objects, atom masks of the graphs, communicator)
    import gc
    reps, stats = 5, {}
    gc.collect()
    gc.disable()                                               # 5 000 graphs: a collection in the timed region costs more than the matrix
    fence()
    t1 = time.perf_counter()
    for _ in range(reps):
        labels = compare_matrix(case["reference"], graphs, stats=stats, **how)
    seconds = all_max(time.perf_counter() - t1) / reps
    gc.enable()
    fence()
    pairs = len(graphs) * (len(graphs) - 1) // 2
    return {"variants": len(graphs), "pairs": pairs, "n_gpus": world, "relations_per_s": round(pairs / seconds, 1),
            "relations_per_s_end_to_end": round(pairs / (seconds + t_graphs + t_columns), 1),
            "end_to_end_covers": "supremal graphs of the variants (from_variant_batch) + their per-variant columns + one matrix",
            "pairs_reaching_dp_on_rank0": stats["dp_pairs"] // reps, "supremal_graphs_s": round(t_graphs, 4),
            "supremal_graphs_s_of_three_calls": [round(t, 4) for t in t_calls],
            "per_variant_columns_s": round(t_columns, 4),
            "matrix_s_covers": "tiles + pre-filters, undecided pairs decided on the device, all-gather, assembly, the full matrix in (pinned) host memory; graphs and their per-variant columns are made before it",
            "matrix_s": round(seconds, 4), "label_histogram": np.bincount(labels.ravel(), minlength=5).tolist(),
            "upper_histogram": np.bincount(labels[np.triu_indices(len(graphs), 1)], minlength=5).tolist(),
            "stages_ms_rank0": {k[8:]: round(v * 1e3 / reps, 2) for k, v in stats.items() if k.startswith("seconds_")}}


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start N fresh rank processes (one per GPU) with the rendezvous in
    their environment, before this process has touched the GPU, and wait for them.  Rank 0 prints the JSON line;
    returns the worst exit code."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    children = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        children.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    worst = 0
    for c in children:
        rc = c.wait()
        if rc != 0 and worst == 0:
            worst = rc if rc > 0 else 128 - rc
    return worst


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--pairs", type=int, default=100000)
    ap.add_argument("--in-flight", type=int, default=3, help="batches in flight per GPU, one stream (context) each: step i is batch "
                                                             "i mod F; 1 = one stream, every step behind the one before")
    ap.add_argument("--preheat-ms", type=float, default=100.0, help="untimed steps before the warm-up, to bring the clocks up")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the secondary configs (graphs, relation matrix)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse the "
                                                      "multi-rank path on a box with fewer GPUs than ranks)")
    ap.add_argument("--share-device", action="store_true", help="rehearsal: all ranks use GPU 0")
    ap.add_argument("--matrix-rccl", action="store_true", help="the relation matrix's all-gather through the library's RCCL "
                                                               "communicator whatever --backend is (the default with nccl)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args.gpus))               # no launcher around us: this process becomes one
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    os.environ.setdefault("MALCS_DEVICE", str(local_rank))

    import torch
    if args.share_device:
        local_rank = 0
        os.environ["MALCS_DEVICE"] = "0"
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)
    reduce_device = "cuda" if args.backend == "nccl" else "cpu"

    import gen
    from algebra_amd import _native

    ctx = _native.Context(local_rank)
    name, cus, mhz = ctx.device_info()
    data = gen.c2_batch(args.pairs, seed=2 + rank)
    syms, a_off, a_len, b_off, b_len = data
    cells = float((a_len.astype(np.float64) * b_len.astype(np.float64)).sum())
    batch = _native.Batch(ctx, syms, a_off, a_len, b_off, b_len, 4)
    # A stream of batches: F of them are resident and in flight at a time, each on a stream (context) of its own, so
    # that the plan of one (three short launches bound by trips to memory) runs beside the distance kernels of another
    # (bound by the VALU).  Step i is one pass over batch i mod F; all K passes are done inside the timed region.
    F = max(1, args.in_flight)
    ctxs, batches, batch_cells = [ctx], [batch], [cells]
    for k in range(1, F):
        ctxs.append(_native.Context(local_rank))
        more = gen.c2_batch(args.pairs, seed=2 + rank + 1000 * k)
        batch_cells.append(float((more[2].astype(np.float64) * more[4].astype(np.float64)).sum()))
        batches.append(_native.Batch(ctxs[k], *more, 4))

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    def all_max(x):
        if dist is None:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=reduce_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    # The card idles at low clocks and takes tens of milliseconds of work to leave them (a launch measured over 50 cold
    # repetitions is 12 % slower than over 1 000): `--preheat-ms` of untimed steps come before the W warm-up steps, so
    # that a short run (--steps 20) measures the same machine state as a long one.
    def step(i=0):
        b = batches[i % F]
        b.replan()                                 # everything that depends on the sequences, queued on the device
        b.launch_distances()

    t_heat = time.perf_counter()
    while (time.perf_counter() - t_heat) * 1e3 < args.preheat_ms:
        for i in range(20):
            step(i)
        torch.cuda.synchronize()
    for i in range(args.warmup):
        step(i)
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    torch.cuda.synchronize()                       # this rank's K steps are done (every stream of the device) ...
    elapsed = time.perf_counter() - t0
    fence()                                        # ... and the job is done when the slowest rank is: MAX below
    cells_done = sum(batch_cells[i % F] for i in range(args.steps))      # what this rank's K steps computed
    if dist is not None:
        elapsed = all_max(elapsed)
        c = torch.tensor([cells_done], dtype=torch.float64, device=reduce_device)
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        total_cells = float(c.item()) / args.steps
    else:
        total_cells = cells_done / args.steps
    # the same K steps on ONE stream (every step behind the one before): what the plan and the kernels cost in a row
    one_ms = elapsed / args.steps * 1e3
    if F > 1:
        for _ in range(args.warmup):
            step(0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step(0)
        torch.cuda.synchronize()
        one_ms = (time.perf_counter() - t0) / args.steps * 1e3

    # a checksum of the results so that a wrong kernel cannot hide behind a fast one
    d = batch.fetch_distances()
    mean_distance = float(np.mean([b.fetch_distances().mean() for b in batches]))

    # dominant kernel: HIP events on the library's stream around `steps` launches of that kernel alone.
    # Its work is priced on the cells it computes: the window class evaluates 32 * limbs rows of every
    # column between the common ends of a pair, not the whole matrix (the full-matrix-equivalent figure is
    # `value`, as the metric says).
    plan = batch.plan_info()
    # (at least 500 launches each, whatever K is: 20 launches of 20 us are too few for an average)
    launches = max(args.steps, 500)
    step_ms = batch.time_distances(launches) / launches           # the launch alone, replayed over the rows of the last plan
    batch.time_dominant_kernel(100)
    back_to_back_ms = batch.time_dominant_kernel(launches) / launches     # one event pair around all launches: + the gap between two kernels
    kernel_timing = {"back_to_back_ms": round(back_to_back_ms, 4)}
    if plan["window_pairs"]:
        # the kernel's average launch duration: one HIP event pair per launch (hipExtLaunchKernelGGL's start / stop events,
        # the dispatch's own time stamps -- what rocprofv3 lists as the kernel's duration), launches back to back
        each = batch.kernel_durations(launches)
        kernel_ms = float(each.mean())
        kernel_timing.update({"how": "mean over one HIP event pair per launch (start/stop events of the dispatch), launches back to back; "
                                     "back_to_back_ms = one event pair around all launches / their number (adds the gap between two kernels of a stream)",
                              "median_ms": round(float(np.median(each)), 4), "min_ms": round(float(each.min()), 4),
                              "p90_ms": round(float(np.percentile(each, 90)), 4)})
        # The kernel as a STEP runs it comes behind a layout launch whose 51 MB of rows are still on their way to memory,
        # and is slower than its replay.  No per-kernel events here (the event packets themselves cost a step 2 us): the
        # plan's three launches alone, timed like the headline, and what a step adds to them.
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(launches):
            batch.replan()
        ctx.sync()
        plan_ms = (time.perf_counter() - t0) / launches * 1e3
        in_step = {"plan_ms": round(plan_ms, 4),
                   "distance_kernels_ms": round(one_ms - plan_ms, 4),
                   "distance_kernels_replayed_ms": round(step_ms, 4),
                   "what": "plan_ms = the plan's three launches alone (same loop as the headline without the launch); distance_kernels_ms = "
                           "one_in_flight.ms - plan_ms = the window kernel + the retry list kernel as a step runs them; replayed: the same two over "
                           "the rows of a plan that is already there (the roofline's kernel_ms is the window kernel of those replays)"}
    else:
        in_step = None
        kernel_ms = back_to_back_ms
        kernel_timing["how"] = "one HIP event pair around all launches / their number"
    padded_cells = None
    if plan["window_pairs"]:
        # priced on the cells between the common ends of every pair AS THEY ARE; what the kernel really runs (texts grown
        # to whole blocks of 32 columns and to the longest of their tile) is reported beside it
        kernel_cells = float(plan["window_cells_unpadded"])
        padded_cells = float(plan["window_cells"])
        kernel_name = f"k_llcs_diag<{plan['window_limbs']},2,true>"
        rows = (f"64 rows (two-limb mode, {plan['two_limb_tiles']} of {plan['window_tiles']} tiles; the others "
                f"{32 * plan['window_limbs']})" if plan["two_limb_tiles"] else f"{32 * plan['window_limbs']} rows")
        algorithmic = (f"{OPS_PER_CELL} lane-op/cell x {kernel_cells:.4g} window cells/launch "
                       f"({rows} x text columns between the pair's common ends, unpadded, {plan['window_pairs']} pairs; "
                       f"as laid out, grown to whole blocks and to the tile's longest: {padded_cells:.4g}; "
                       f"{cells:.4g} full-matrix-equivalent cells)")
    else:
        kernel_cells = cells
        kernel_name = "k_llcs_bitpar<4,2,true>"
        algorithmic = f"{OPS_PER_CELL} lane-op/cell x {cells:.4g} cells/launch"
    achieved = kernel_cells * OPS_PER_CELL / (kernel_ms * 1e-3) / 1e12

    # Beside the headline (rank 0's figures; every rank does the same work):
    #   resident_replay the launch alone over the rows of a plan that is already there (the figure of the earlier rounds'
    #                   headline: what a caller gets who computes the SAME batch again and again);
    #   end_to_end      from the caller's host buffers (one byte per symbol, pageable numpy arrays) to distances in host
    #                   memory: malcs_batch_create (planes packed on the host's cores, copied packed, classes from the
    #                   lengths, plan) + launch + fetch + destroy.  PCIe-inclusive: never `value`.
    reps = 20
    assert np.array_equal(batch.fetch_distances(), d)
    for _ in range(2):
        _native.Batch(ctx, syms, a_off, a_len, b_off, b_len, 4).close()
    t0 = time.perf_counter()
    for _ in range(reps):
        once = _native.Batch(ctx, syms, a_off, a_len, b_off, b_len, 4)
        d_once = once.distances()
        once.close()
    e2e_s = (time.perf_counter() - t0) / reps
    assert np.array_equal(d_once, d)

    # the same batches from sequences that are bit planes already (malcs_batch_create_packed): a caller that keeps its
    # windows packed -- in host memory, or in HBM -- skips the pack of 100 MB of bytes per batch
    from util import pack_arena
    planes, where = pack_arena(syms, np.concatenate([a_off, b_off]), np.concatenate([a_len, b_len]), 2)
    pa, pb = where[:args.pairs].copy(), where[args.pairs:].copy()
    resident = torch.from_numpy(planes.view(np.int32)).cuda()
    torch.cuda.synchronize()
    packed_s = {}
    for name_, arena, n_dw in (("host_planes", planes, None), ("device_planes", int(resident.data_ptr()), planes.size)):
        for timed in (False, True):
            t0 = time.perf_counter()
            for _ in range(reps if timed else 1):
                once = _native.Batch.from_packed(ctx, arena, pa, a_len, pb, b_len, 2, n_dwords=n_dw)
                d_once = once.distances()
                once.close()
            packed_s[name_] = (time.perf_counter() - t0) / reps
        assert np.array_equal(d_once, d)
    del resident

    # the relation matrix is a job of all ranks (sharded tiles + all-gather): every rank takes part.  It runs beside the
    # headline, which is measured by now: a failure or a hang there (the RCCL path has only ever run on one card before
    # the driver's scaling run) is reported in the line instead of costing it.
    matrix = None
    hung = False
    if not args.no_extra:
        import threading
        box = {}

        def run_matrix():
            try:
                comm = gather = None
                if world > 1 and (args.backend == "nccl" or args.matrix_rccl):
                    # the library's own RCCL communicator (no torch in that path); its 128-byte id travels over the
                    # process group this script has anyway
                    where = "cuda" if args.backend == "nccl" else "cpu"
                    uid = torch.zeros(129, dtype=torch.uint8, device=where)    # 128 bytes of id + "rank 0 has one"
                    failure = None
                    if rank == 0:
                        try:
                            uid.copy_(torch.frombuffer(bytearray(_native.comm_unique_id() + b"\x01"), dtype=torch.uint8))
                        except Exception as e:                     # noqa: BLE001 -- the other ranks wait in the broadcast
                            failure = e
                    dist.broadcast(uid, 0)
                    if failure is not None:
                        raise failure
                    raw = bytes(uid.cpu().numpy().tobytes())
                    if raw[128] != 1:
                        raise RuntimeError("rank 0 could not make a communicator id")
                    comm = _native.Comm(ctx, raw[:128], rank, world)
                elif world > 1:
                    def gather(block):                             # rehearsal on one card: gloo carries the label blocks
                        send = torch.from_numpy(np.ascontiguousarray(block))
                        recv = torch.empty((world * block.shape[0],) + tuple(block.shape[1:]), dtype=torch.uint8)
                        dist.all_gather_into_tensor(recv, send)
                        return recv.numpy()
                box["matrix"] = relation_matrix(ctx, gen, rank, world, comm, gather, fence, all_max)
                if comm is not None:
                    comm.close()
            except BaseException as e:                             # noqa: BLE001 -- reported, not raised
                box["error"] = f"{type(e).__name__}: {e}"

        if world == 1:
            run_matrix()
        else:
            torch.cuda.set_device(local_rank)
            worker = threading.Thread(target=lambda: (torch.cuda.set_device(local_rank), run_matrix()), daemon=True)
            worker.start()
            worker.join(float(os.environ.get("BENCH_MATRIX_TIMEOUT", "150")))
            hung = worker.is_alive()
            if hung:
                box["error"] = "relation matrix did not finish in time (a rank failed or the all-gather hangs)"
        matrix = box.get("matrix")
        matrix_error = box.get("error")
    else:
        matrix_error = None

    if rank == 0:
        # HBM bytes per launch of that kernel: NOT measured in this run (counters need rocprofv3 passes of their own);
        # read from the committed PMC summary of the same command and tagged with its source
        traffic, traffic_source = None, None
        for tag in ("r03", "r02", "r01"):
            pmc = os.path.join(ROOT, "profiles", f"{tag}_pmc.json")
            if os.path.exists(pmc) and args.pairs == 100000:
                with open(pmc) as f:
                    traffic = json.load(f)["kernels"].get(kernel_name, {}).get("hbm_bytes_per_launch")
                if traffic is not None:
                    traffic_source = f"profiles/{tag}_pmc.json"
                    break
        # What a stream of batches in flight asks of memory: the bytes all five launches of a step move (same committed
        # counter passes, a launch each) against the time a batch takes.
        step_bytes = None
        if traffic_source is not None:
            with open(os.path.join(ROOT, traffic_source)) as f:
                per_kernel = {k: v.get("hbm_bytes_per_launch") for k, v in json.load(f)["kernels"].items()
                              if k.startswith(("k_plan_", "k_diag_layout", "k_llcs_diag", "k_llcs_bitpar_list"))}
            if per_kernel and all(v is not None for v in per_kernel.values()):
                step_bytes = {"per_kernel": per_kernel, "per_batch": int(sum(per_kernel.values())),
                              "GB_per_s": round(sum(per_kernel.values()) * args.steps / (elapsed / 1.0) / 1e9, 1),
                              "frac_of_8000": round(sum(per_kernel.values()) * args.steps / elapsed / 8e12, 4),
                              "what": "HBM bytes of the plan's three launches + the two distance kernels of one batch (" + traffic_source +
                                      ") over this run's time per batch on rank 0: what a stream of batches in flight asks of the memory system"}
        replay_s = step_ms * 1e-3
        out = {
            "metric": "GCUPS (batched LCS edit distance, 1 kb windows)",
            "value": round(total_cells * args.steps / elapsed / 1e9, 1),
            "unit": "GCUPS",
            "pairs_per_s": round(args.pairs * world * args.steps / elapsed, 1),
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u32", "data": "synthetic",
            "config": {"workload": f"C2: batches of {args.pairs} pairs/GPU, 1 kb reference windows, ~2% indels (seeds 2+rank+1000k); resident in HBM "
                                   "when a step starts: the packed bit planes (2 bit/base) and the per-pair offset / length "
                                   "tables, nothing derived from the sequences; a step = one batch's plan on the device "
                                   "(common ends, sort into tiles, sample, pieces, rows) + its distance kernels; "
                                   f"{F} batch(es) in flight per GPU, a stream each (step i is batch i mod {F})",
                       "pairs_per_gpu": args.pairs, "batches_in_flight": F,
                       "parallelism": f"independent pairs x {world} rank(s)", "device": name},
            "mean_distance": round(mean_distance, 3),
            "roofline": {"bound": "valu", "achieved": round(achieved, 2), "peak": round(PEAK_LANE_OPS / 1e12, 1),
                         "unit": "T int32 lane-op/s", "frac": round(achieved * 1e12 / PEAK_LANE_OPS, 4),
                         "traffic": traffic, "traffic_source": traffic_source, "kernel": kernel_name, "kernel_ms": round(kernel_ms, 4),
                         "kernel_timing": kernel_timing,
                         "in_step": in_step, "step_traffic": step_bytes,
                         "frac_back_to_back": round(kernel_cells * OPS_PER_CELL / (back_to_back_ms * 1e-3) / PEAK_LANE_OPS, 4),
                         "launches_timed": launches, "algorithmic": algorithmic,
                         "cells_as_laid_out": padded_cells,
                         "frac_on_cells_as_laid_out": None if padded_cells is None else round(padded_cells * OPS_PER_CELL / (kernel_ms * 1e-3) / PEAK_LANE_OPS, 4)},
            "plan": plan,
            "plan_inclusive": {"pairs_per_s": round(args.pairs * world * args.steps / elapsed, 1), "ms": round(elapsed / args.steps * 1e3, 4),
                               "what": "= the headline: per batch, plan on the device + launch, nothing copied back"},
            "one_in_flight": {"pairs_per_s": round(args.pairs / (one_ms * 1e-3), 1), "ms": round(one_ms, 4),
                              "what": "rank 0, the same steps on one stream, every step behind the one before (rounds 1-3's headline until "
                                      "the batches went in flight side by side): the plan's three launches and the distance kernels in a row"},
            "resident_replay": {"pairs_per_s": round(args.pairs / replay_s, 1), "ms": round(step_ms, 4),
                                "GCUPS": round(cells / replay_s / 1e9, 1),
                                "what": "the distance kernels alone, replayed over the rows of a plan that is already there (HIP events, "
                                        f"{launches} launches): the earlier rounds' headline, not what a caller of one batch gets"},
            "end_to_end": {"pairs_per_s": round(args.pairs / e2e_s, 1), "ms": round(e2e_s * 1e3, 3),
                           "what": f"per batch, from {syms.nbytes / 1e6:.0f} MB of host bytes (pageable) to distances in host memory: "
                                   "create (pack on host cores, copy packed, plan) + launch + fetch + destroy"},
            "end_to_end_packed": {
                "what": "as end_to_end, but the caller's sequences are bit planes already (malcs_batch_create_packed, "
                        f"{planes.nbytes / 1e6:.0f} MB): create (copy the planes, plan) + launch + fetch + destroy",
                "from_host_planes": {"pairs_per_s": round(args.pairs / packed_s["host_planes"], 1), "ms": round(packed_s["host_planes"] * 1e3, 3)},
                "from_device_planes": {"pairs_per_s": round(args.pairs / packed_s["device_planes"], 1), "ms": round(packed_s["device_planes"] * 1e3, 3)}},
        }
        if matrix is not None:
            out["relations_per_s"] = matrix["relations_per_s"]
            out["extra"] = {"c4_relation_matrix": matrix}
        elif matrix_error is not None:
            out["extra"] = {"c4_relation_matrix": {"error": matrix_error}}
        if world == 1 and not args.no_extra and matrix is not None:
            out["extra"].update(extra_configs(ctx, _native, gen))
            try:
                out["extra"].update(long_window_configs(ctx, _native, gen))
            except Exception as e:                                  # noqa: BLE001 -- reported beside the headline, not instead of it
                out["extra"]["c3_c5_error"] = f"{type(e).__name__}: {e}"
        if world == 1 and not args.no_cpu_baseline:                # rank 0 at N = 1 only
            out["cpu_baseline"] = cpu_baseline(data, cells / args.pairs)
            out["end_to_end"]["vs_cpu_port"] = round(out["end_to_end"]["pairs_per_s"] / out["cpu_baseline"]["pairs_per_s"], 2)
        print(json.dumps(out), flush=True)

    if hung or (world > 1 and matrix_error is not None):
        # the headline is printed; the job FAILED all the same (3: the relation matrix raised, 4: it did not come back) --
        # no clean shutdown: a communicator that hangs may never return
        sys.stdout.flush()
        sys.stderr.write(f"bench.py: relation matrix failed on rank {rank}: {matrix_error}\n")
        sys.stderr.flush()
        os._exit(4 if hung else 3)
    if matrix_error is not None:
        sys.stderr.write(f"bench.py: relation matrix failed: {matrix_error}\n")
        for b in batches: b.close()
        for c in ctxs: c.close()
        sys.exit(3)
    for b in batches: b.close()
    for c in ctxs: c.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
