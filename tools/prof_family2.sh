# Kernel trace and HBM counters of family 2's bit-vector kernels at the bench's size (run on the GPU box:
# gpurun -- bash tools/prof_family2.sh [TAG]); summaries land in profiles/<TAG>_family2_* and gpurun_out/prof_<TAG>/.
set -e
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp
R=/root/repo
P=$R/gpurun_out/raw_prof2
rm -rf $P; mkdir -p $P $R/gpurun_out/prof_$TAG
B="python3 $R/tools/prof_family2.py"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $P/stats -o stats -- $B > $P/stats.log 2>&1
REPS=1 timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE -d $P/fetch -o fetch -- $B > $P/fetch.log 2>&1
REPS=1 timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE -d $P/write -o write -- $B > $P/write.log 2>&1
cd $R
python3 tools/prof_summary.py ${TAG}_family2 $P/stats $P/fetch $P/write "$B (100 000 C2 pairs through k_lcs_forward / k_lcs_reach, MALCS_SWEEP_DMAX=-1; stats: 3 calls, counters: 1 call, a pass each)" > /dev/null
grep "tie bytes" $P/stats.log | tail -1 > gpurun_out/prof_$TAG/family2_run.txt
cp profiles/${TAG}_family2_bench_stats.txt profiles/${TAG}_family2_pmc.json gpurun_out/prof_$TAG/
rm -rf $P
head -8 profiles/${TAG}_family2_bench_stats.txt; cat gpurun_out/prof_$TAG/family2_run.txt
