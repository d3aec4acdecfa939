"""Print average PMC counter values per kernel from a rocprofv3 --pmc result directory (dev tool)."""
import glob, sqlite3, sys
for path in glob.glob(sys.argv[1] + "/**/*.db", recursive=True):
    db = sqlite3.connect(path)
    like = sys.argv[2] if len(sys.argv) > 2 else "%"
    rows = db.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection "
                      "where kernel_name like ? group by kernel_name, counter_name", (like,)).fetchall()
    for k, c, v, n in rows:
        print(f"{k[:60]:60s} {c:28s} {v:16.1f} n={n}")
