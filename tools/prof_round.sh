# The round's profile passes over the bench (run on the GPU box: gpurun -- bash tools/prof_round.sh [TAG]); the summaries
# are written on the box (the raw result databases of the counter passes are tens of megabytes) and come back under
# gpurun_out/prof_<TAG>/: copy them into profiles/.
set -e
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp
R=/root/repo
P=$R/gpurun_out/raw_prof
rm -rf $P $R/gpurun_out/prof_$TAG; mkdir -p $P $R/gpurun_out/prof_$TAG
B="python3 $R/bench.py --no-cpu-baseline --no-extra --in-flight 1"   # one stream: kernels of batches in flight side by side stretch each other, and the counters serialise them anyway
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $P/stats -o stats -- $B --steps 200 --warmup 20 > $P/stats.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE -d $P/fetch -o fetch -- $B --steps 20 --warmup 2 > $P/fetch.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE -d $P/write -o write -- $B --steps 20 --warmup 2 > $P/write.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $P/valu -o valu -- $B --steps 20 --warmup 2 > $P/valu.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY -d $P/lds -o lds -- $B --steps 20 --warmup 2 > $P/lds.log 2>&1
cd $R
# the same trace of the bench as the driver runs it (three batches in flight: their kernels overlap and stretch each other;
# the replays the roofline's kernel_ms comes from are on one stream there too)
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $P/stats3 -o stats3 -- python3 $R/bench.py --no-cpu-baseline --no-extra --steps 200 --warmup 20 > $P/stats3.log 2>&1
# C2's 100 000 graphs by the default path: one-wave sweeps with several pairs per wave, run ordering, construction
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $P/graphs -o graphs -- python3 $R/tools/time_gbuild.py > $P/graphs.log 2>&1
cd $R
python3 tools/prof_summary.py ${TAG}_graphs $P/graphs > /dev/null
sed -i "1a # command: rocprofv3 --kernel-trace --stats -- python3 tools/time_gbuild.py   (100 000 C2 pairs, Batch.graphs() three times; extra.c2_lcs_graphs of the bench)" profiles/${TAG}_graphs_bench_stats.txt
python3 tools/prof_summary.py ${TAG}_in_flight $P/stats3 > /dev/null
sed -i "1a # command: rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --no-extra --steps 200 --warmup 20   (the default: three batches in flight)" profiles/${TAG}_in_flight_bench_stats.txt
python3 tools/prof_summary.py $TAG $P/stats $P/fetch $P/write "$B --steps 200 --warmup 20 (stats) / --steps 20 --warmup 2 (FETCH_SIZE and WRITE_SIZE, a pass each)" > /dev/null
python3 tools/valu_summary.py $TAG $P/valu "rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -- $B --steps 20 --warmup 2" > /dev/null
python3 - $TAG $P/lds "$B" <<'PY'
import glob, json, sqlite3, sys
tag, path, cmd = sys.argv[1:4]
db = sqlite3.connect(glob.glob(path + "/**/*.db", recursive=True)[0])
rows = db.execute("select counter_name, avg(value) from counters_collection where kernel_name like '%k_llcs_diag%' group by counter_name").fetchall()
json.dump({"command": "rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY -- " + cmd + " --steps 20 --warmup 2",
           "kernel": "k_llcs_diag<3,2,true>", "per_launch": dict(rows)}, open(f"profiles/{tag}_lds.json", "w"), indent=1)
PY
sed -i "1a # command: rocprofv3 --kernel-trace --stats -- $B --steps 200 --warmup 20" profiles/${TAG}_bench_stats.txt
cp profiles/${TAG}_graphs_bench_stats.txt profiles/${TAG}_in_flight_bench_stats.txt profiles/${TAG}_bench_stats.txt profiles/${TAG}_pmc.json profiles/${TAG}_valu.json profiles/${TAG}_lds.json gpurun_out/prof_$TAG/
grep "^{" $P/stats.log > gpurun_out/prof_$TAG/bench_line_under_stats.json || true
rm -rf $P
head -6 profiles/${TAG}_bench_stats.txt
