# Kernel trace of plan + launch per batch (run on the GPU box: bash tools/prof_plan.sh [OUT]); the per-kernel table lands in gpurun_out/
OUT=${1:-gpurun_out/r3/plan_kernels.csv}
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pp
timeout -k 10 150 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -o pp -- python3 $R/tools/prof_plan.py > /tmp/pp.log 2>&1 || { tail -5 /tmp/pp.log; exit 1; }
cd $R
f=$(find /tmp/pp -name "*kernel_stats.csv" | head -1)
python3 - "$f" "$OUT" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
with open(sys.argv[2], "w") as out:
    for r in rows:
        line = "%-70s calls %6s avg %9.2f us min %9.2f max %9.2f  %5.1f %%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, float(r["Percentage"]))
        print(line); out.write(line + "\n")
PY
# the last plan + launch as a timeline: start and end of every kernel relative to the first
t=$(find /tmp/pp -name "*kernel_trace.csv" | head -1)
python3 - "$t" "${OUT%.csv}_timeline.txt" <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
ends = [i for i, r in enumerate(rows) if "k_plan_ends" in r["Kernel_Name"]]
seg = rows[ends[-2]:ends[-1]] if len(ends) >= 2 else rows[-12:]
t0 = int(seg[0]["Start_Timestamp"])
with open(sys.argv[2], "w") as out:
    for r in seg:
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:40]
        line = "%-40s start %8.2f end %8.2f us  (%6.2f)  queue %s" % (name, (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Queue_Id", "?"))
        print(line); out.write(line + "\n")
PY
