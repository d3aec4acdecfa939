"""Summarise rocprofv3 result databases into profiles/ (text + JSON).

    python tools/prof_summary.py TAG STATS_DIR [FETCH_DIR WRITE_DIR [COMMAND]]

STATS_DIR: output of `rocprofv3 --kernel-trace --stats`, FETCH_DIR / WRITE_DIR: separate `--pmc FETCH_SIZE` /
`--pmc WRITE_SIZE` passes over the same command.  gfx950 correction (MI355X_MICROARCH.md, HBM section):
FETCH_SIZE tallies 128-byte read requests at 64 bytes, so fetched bytes = 2 x FETCH_SIZE (KB); WRITE_SIZE is exact.
"""
import glob, json, re, sqlite3, sys


def db_of(path):
    return sqlite3.connect(glob.glob(path + "/**/*.db", recursive=True)[0])


def short(name):
    m = re.search(r"(k_\w+<[^>]*>|k_\w+)", name)
    return m.group(1).replace(" ", "") if m else name[:60]


def stats(path):
    rows = db_of(path).execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) "
                               "from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    lines = ["# rocprofv3 --kernel-trace --stats summary (durations in ns)",
             f"{'calls':>6} {'total_ns':>14} {'avg_ns':>12} {'min_ns':>10} {'max_ns':>10} {'pct':>6}  kernel"]
    for name, n, tot, avg, mn, mx in rows:
        lines.append(f"{n:6d} {tot:14.0f} {avg:12.1f} {mn:10.0f} {mx:10.0f} {100*tot/total:6.2f}  {short(name)}")
    # the window kernel by what ran before it on the device: right behind the layout (a step of the bench) or behind its
    # own kind (the replays the roofline's kernel_ms is measured on)
    try:
        seq = db_of(path).execute("select name, duration from kernels order by start").fetchall()
    except sqlite3.Error:
        seq = []
    split = {}
    for (prev, _), (name, dur) in zip(seq, seq[1:]):
        if "k_llcs_diag" in name:
            key = "behind k_diag_layout (in a step)" if "k_diag_layout" in prev else "behind its own kind (replayed)" if "k_llcs_" in prev else "behind anything else"
            split.setdefault((short(name), key), []).append(dur)
    if split:
        lines.append("# the window kernel by its predecessor on the device")
        for (name, key), d in sorted(split.items()):
            d.sort()
            lines.append(f"{len(d):6d} {sum(d):14.0f} {sum(d)/len(d):12.1f} {d[0]:10.0f} {d[-1]:10.0f} {'':>6}  {name} {key}; median {d[len(d)//2]:.0f}")
    return "\n".join(lines) + "\n"


def counter(path, name):
    """Per kernel: (grid, average counter value, launches, average duration) over the launches with the kernel's
    LARGEST grid -- the batch the bench times; plans also run the same kernels on a sample or on a few pairs."""
    out = {}
    db = db_of(path)
    for k, grid in db.execute("select kernel_name, max(grid_size) from counters_collection where counter_name = ? "
                              "group by kernel_name", (name,)).fetchall():
        v, n, dur = db.execute("select avg(value), count(*), avg(duration) from counters_collection where counter_name = ? "
                               "and kernel_name = ? and grid_size = ?", (name, k, grid)).fetchone()
        out[short(k)] = (grid, v, n, dur)
    return out


if __name__ == "__main__":
    tag, stats_dir = sys.argv[1], sys.argv[2]
    text = stats(stats_dir)
    open(f"profiles/{tag}_bench_stats.txt", "w").write(text)
    print(text)
    if len(sys.argv) > 4:
        fetch, write = counter(sys.argv[3], "FETCH_SIZE"), counter(sys.argv[4], "WRITE_SIZE")
        rec = {"command": sys.argv[5] if len(sys.argv) > 5 else
               "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline (separate passes)",
               "correction": "gfx950 FETCH_SIZE tallies 128-B read requests at 64 B (MI355X_MICROARCH.md, HBM): fetched bytes = 2 x FETCH_SIZE; WRITE_SIZE is exact",
               "kernels": {}}
        for k in fetch:
            if k not in write or not k.startswith("k_"):
                continue
            grid, f, n, dur = fetch[k]
            w = write[k][1]
            rec["kernels"][k] = {"grid_threads": grid, "launches_sampled": n, "FETCH_SIZE_KB_per_launch": f,
                                 "WRITE_SIZE_KB_per_launch": w, "hbm_bytes_per_launch": int(2 * f * 1024 + w * 1024),
                                 "avg_duration_ns_under_pmc": dur}
        json.dump(rec, open(f"profiles/{tag}_pmc.json", "w"), indent=1)
        print(json.dumps(rec, indent=1))
