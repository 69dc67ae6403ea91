set -o pipefail
mkdir -p gpurun_out/r1f && cd /root/repo
timeout -k 10 900 python -m pytest tests -q -m gpu 2>&1 | tail -3 > gpurun_out/r1f/tests.log; cat gpurun_out/r1f/tests.log
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r1f/bench.log 2>&1 && tail -1 gpurun_out/r1f/bench.log | cut -c1-160
timeout -k 10 300 python bench.py --mode frozen --no-cpu-baseline > gpurun_out/r1f/bench_frozen.log 2>&1 && tail -1 gpurun_out/r1f/bench_frozen.log | cut -c60-160
timeout -k 10 300 python bench.py --dtype bf16 --no-cpu-baseline > gpurun_out/r1f/bench_bf16.log 2>&1 && tail -1 gpurun_out/r1f/bench_bf16.log | cut -c60-160
timeout -k 10 300 python bench.py --model cnn32 --steps 50 --warmup 10 > gpurun_out/r1f/bench_cnn32.log 2>&1 && tail -1 gpurun_out/r1f/bench_cnn32.log | cut -c60-160
timeout -k 10 300 python bench.py --model wrn --steps 10 --warmup 3 > gpurun_out/r1f/bench_wrn.log 2>&1 && tail -1 gpurun_out/r1f/bench_wrn.log | cut -c60-160
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r1f/stats -o run -- python $R/bench.py --steps 8 --warmup 5 --no-cpu-baseline > $R/gpurun_out/r1f/stats.log 2>&1; echo stats rc=$?
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r1f/stats_wrn -o run -- python $R/bench.py --model wrn --steps 6 --warmup 3 --no-cpu-baseline > $R/gpurun_out/r1f/stats_wrn.log 2>&1; echo stats_wrn rc=$?
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/r1f/pmc_fetch -o run -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline > $R/gpurun_out/r1f/pmc_fetch.log 2>&1; echo fetch rc=$?
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/r1f/pmc_write -o run -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline > $R/gpurun_out/r1f/pmc_write.log 2>&1; echo write rc=$?
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $R/gpurun_out/r1f/pmc_sq -o run -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline > $R/gpurun_out/r1f/pmc_sq.log 2>&1; echo sq rc=$?
ls $R/gpurun_out/r1f
