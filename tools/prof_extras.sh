# Kernel trace of the whole bench, secondary configs included (graphs, bit-vector path, relation matrix): run on the GPU
# box (gpurun -- bash tools/prof_extras.sh [TAG]); the summary comes back as gpurun_out/prof_<TAG>/<TAG>_bench_extras_stats.txt
set -e
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp
R=/root/repo
P=$R/gpurun_out/raw_prof_extras
rm -rf $P; mkdir -p $P $R/gpurun_out/prof_$TAG
CMD="python3 $R/bench.py --no-cpu-baseline --steps 200 --warmup 20"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $P/stats -o stats -- $CMD > $P/stats.log 2>&1
cd $R
python3 - $TAG $P/stats "$CMD" <<'PY'
import sys
sys.path.insert(0, "tools")
from prof_summary import stats
tag, path, cmd = sys.argv[1:4]
text = "# rocprofv3 --kernel-trace --stats -- " + cmd.replace("/root/repo/", "") + " (secondary configs included)\n" + stats(path)
open(f"gpurun_out/prof_{tag}/{tag}_bench_extras_stats.txt", "w").write(text)
print("\n".join(text.splitlines()[:14]))
PY
grep "^{" $P/stats.log > gpurun_out/prof_$TAG/bench_line_under_extras_stats.json || true
rm -rf $P
