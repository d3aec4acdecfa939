set -e
cd /tmp && export TMPDIR=/tmp
R=/root/repo
rm -rf $R/gpurun_out/p2_*
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/p2_stats -o stats -- python3 $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extra > $R/gpurun_out/p2_stats.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/p2_fetch -o fetch -- python3 $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-extra > $R/gpurun_out/p2_fetch.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/p2_write -o write -- python3 $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-extra > $R/gpurun_out/p2_write.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $R/gpurun_out/p2_valu -o valu -- python3 $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-extra > $R/gpurun_out/p2_valu.log 2>&1
tail -1 $R/gpurun_out/p2_stats.log | cut -c1-400
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY -d $R/gpurun_out/p2_lds -o lds -- python3 $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-extra > $R/gpurun_out/p2_lds.log 2>&1
