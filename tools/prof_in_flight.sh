# Kernel trace of three batches in flight (run on the GPU box: bash tools/prof_in_flight.sh); prints 40 kernels of the steady state
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pf
timeout -k 10 150 rocprofv3 --kernel-trace --output-format csv -d /tmp/pf -o pf -- python3 $R/tools/time_two_in_flight.py 300 > /tmp/pf.log 2>&1 || { tail -5 /tmp/pf.log; exit 1; }
cd $R
t=$(find /tmp/pf -name "*kernel_trace.csv" | head -1)
python3 - "$t" <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-120:-60]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:28]
    print("%-28s q%-3s start %8.2f end %8.2f (%6.2f)" % (name, r.get("Queue_Id", "?"), (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
PY
