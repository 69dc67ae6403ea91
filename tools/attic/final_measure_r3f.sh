set -o pipefail
# after the last block went to its class-token rows: GPU suite, ViT bench lines (full / frozen / eval / bf16), serialised + overlapped kernel tables
O=gpurun_out/r3k; mkdir -p $O; cd /root/repo
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; tail -2 $O/pytest_gpu.log
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err && cut -c1-160 $O/bench.json
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --dtype bf16 --no-cpu-baseline > $O/bench_bf16.json 2> $O/bench_bf16.err && cut -c1-160 $O/bench_bf16.json
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --mode frozen --no-cpu-baseline > $O/bench_frozen.json 2> $O/bench_frozen.err && cut -c1-160 $O/bench_frozen.json
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --mode eval --no-cpu-baseline > $O/bench_eval.json 2> $O/bench_eval.err && cut -c1-160 $O/bench_eval.json
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/$O/st -o run -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --serial-kernels > $R/$O/st.log 2>&1
cd $R; f=$(find $O/st -name "*results.db" | head -1); python tools/pmc_summary.py stats $f > $O/serial_kernel_stats.csv; rm -rf $O/st; head -8 $O/serial_kernel_stats.csv | cut -c1-120
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/$O/st2 -o run -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > $R/$O/st2.log 2>&1
cd $R; f=$(find $O/st2 -name "*results.db" | head -1); python tools/pmc_summary.py stats $f > $O/kernel_stats.csv; rm -rf $O/st2; head -4 $O/kernel_stats.csv | cut -c1-120
