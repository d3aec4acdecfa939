# dev: the layout kernel with 1 / 2 / 4 waves per place (MALCS_LAYOUT_WPP): kernel times of a plan + launch
R=$(pwd); cd /tmp && export TMPDIR=/tmp
for w in ${@:-4 2}; do
  export MALCS_LAYOUT_WPP=$w
  rm -rf /tmp/pl; timeout -k 10 100 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pl -o pl -- python3 $R/tools/prof_plan.py > /tmp/pl.log 2>&1 || { tail -3 /tmp/pl.log; exit 1; }
  python3 - $(find /tmp/pl -name "*kernel_stats.csv") $w <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "k_plan" in r["Name"] or "k_diag_layout" in r["Name"]:
        print("WPP=%s %-40s %8.2f us" % (sys.argv[2], r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:40], float(r["AverageNs"]) / 1e3))
PY
done
