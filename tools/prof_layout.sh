# dev: the layout kernel under settings "WPP:BS" (waves per place, threads per workgroup): kernel times of a plan + launch
R=$(pwd); cd /tmp && export TMPDIR=/tmp
for cfg in ${@:-2:512 4:1024}; do
  export MALCS_LAYOUT_WPP=${cfg%%:*} MALCS_LAYOUT_BS=${cfg##*:}
  rm -rf /tmp/pl; timeout -k 10 100 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pl -o pl -- python3 $R/tools/prof_plan.py > /tmp/pl.log 2>&1 || { tail -3 /tmp/pl.log; exit 1; }
  python3 - $(find /tmp/pl -name "*kernel_stats.csv") $cfg <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "k_plan" in r["Name"] or "k_diag_layout" in r["Name"]:
        print("%-8s %-44s %8.2f us" % (sys.argv[2], r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:44], float(r["AverageNs"]) / 1e3))
PY
done
