set -o pipefail
O=gpurun_out/r3j; mkdir -p $O; cd /root/repo
timeout -k 10 300 python bench.py --model wrn --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_wrn.json 2> $O/bench_wrn.err; cut -c1-160 $O/bench_wrn.json
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; cut -c1-160 $O/bench.json
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/$O/st -o run -- python $R/bench.py --model wrn --steps 6 --warmup 3 --no-cpu-baseline --serial-kernels > $R/$O/st.log 2>&1
cd $R; f=$(find $O/st -name "*results.db" | head -1); python tools/pmc_summary.py stats $f > $O/wrn_serial_kernel_stats.csv; rm -rf $O/st; head -5 $O/wrn_serial_kernel_stats.csv | cut -c1-120
