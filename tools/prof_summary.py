"""Summarise rocprofv3 result databases (gpurun_out/prof/*) into profiles/ (text + JSON)."""
import json, sqlite3, sys

def stats(db_path):
    db = sqlite3.connect(db_path)
    rows = db.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) "
                      "from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    lines = ["# rocprofv3 --kernel-trace --stats summary (durations in ns)",
             f"{'calls':>6} {'total_ns':>14} {'avg_ns':>12} {'min_ns':>10} {'max_ns':>10} {'pct':>6}  kernel"]
    for name, n, tot, avg, mn, mx in rows:
        lines.append(f"{n:6d} {tot:14.0f} {avg:12.1f} {mn:10.0f} {mx:10.0f} {100*tot/total:6.2f}  {name}")
    return "\n".join(lines) + "\n", rows

def counter(db_path, like):
    db = sqlite3.connect(db_path)
    return db.execute("select counter_name, avg(value), count(*) from counters_collection where kernel_name like ? "
                      "group by counter_name", (like,)).fetchall()

if __name__ == "__main__":
    tag, prof = sys.argv[1], sys.argv[2]
    text, rows = stats(f"{prof}/stats/r01_results.db")
    open(f"profiles/{tag}_stats.txt", "w").write(text)
    print(text)
