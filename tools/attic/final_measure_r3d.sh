# refresh of the exact-fp32 (parity-mode) lines and kernel table after the packed-weight / 32-deep-k-tile changes (gpurun_out/r3i/)
set -o pipefail
O=gpurun_out/r3i
mkdir -p $O && cd /root/repo
b() { name=$1; shift; timeout -k 10 300 python bench.py "$@" > $O/$name.json 2> $O/$name.err && cut -c1-160 $O/$name.json; }
b bench_cnn32_parity --model cnn32 --steps 50 --warmup 10 --parity-mode --no-cpu-baseline
b bench_wrn_parity --model wrn --steps 6 --warmup 2 --parity-mode --no-cpu-baseline
b bench_wrn32_parity --model wrn --res 32 --steps 10 --warmup 3 --parity-mode --no-cpu-baseline
b bench --steps 20 --warmup 5
timeout -k 10 300 python tools/parity_conv_bench.py 0 > $O/parity_conv_bench.log 2>&1
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/$O/stats_wrn_parity -o run -- python $R/bench.py --model wrn --steps 4 --warmup 2 --no-cpu-baseline --parity-mode --serial-kernels > $R/$O/stats_wrn_parity.log 2>&1
cd $R
f=$(find $O/stats_wrn_parity -name "*results.db" | head -1); [ -n "$f" ] && python tools/pmc_summary.py stats $f > $O/stats_wrn_parity_kernel_stats.csv
rm -rf $O/stats_wrn_parity
timeout -k 10 600 python -m pytest tests/test_gpu_parity_big.py -q -s > $O/parity_big.log 2>&1; echo parity rc=$?
ls $O
