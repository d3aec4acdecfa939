# dev: the layout kernel under a few settings: time (kernel trace)
R=$(pwd); cd /tmp && export TMPDIR=/tmp
for cfg in "4 0" "2 0" "4 4"; do
  set -- $cfg
  export MALCS_LAYOUT_WPP=$1 MALCS_LAYOUT_NT=$2
  rm -rf /tmp/pl; timeout -k 10 100 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pl -o pl -- python3 $R/tools/prof_plan.py > /tmp/pl.log 2>&1 || { tail -3 /tmp/pl.log; exit 1; }
  t=$(grep k_diag_layout $(find /tmp/pl -name "*kernel_stats.csv") | head -1 | cut -d, -f4)
  echo "WPP=$1 flags=$2: avg ns $t"
done
