"""Total time per HIP API call from a `rocprofv3 --hip-trace` result directory (dev tool)."""
import glob, sqlite3, sys
db = sqlite3.connect(glob.glob(sys.argv[1] + "/**/*.db", recursive=True)[0])
tables = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
view = "regions" if "regions" in tables else next(t for t in tables if "region" in t)
cols = [r[1] for r in db.execute(f"pragma table_info({view})")]
rows = db.execute(f"select name, count(*), sum(end - start), avg(end - start) from {view} group by name order by 3 desc limit 25").fetchall()
for name, n, tot, avg in rows:
    print(f"{n:8d} {tot/1e6:10.2f} ms {avg/1e3:9.1f} us  {name}")
