# round-3 second measurement pass (after the parity-mode convolution work, the fused CBAM unit and the BatchNorm backward change):
# the configurations those touched + the headline for the same box.  Outputs under gpurun_out/r3g/.
set -o pipefail
O=gpurun_out/r3g
mkdir -p $O && cd /root/repo
b() { name=$1; shift; timeout -k 10 300 python bench.py "$@" > $O/$name.json 2> $O/$name.err && cut -c1-160 $O/$name.json; }
b bench --steps 20 --warmup 5
b bench_cnn32 --model cnn32 --steps 50 --warmup 10 --no-cpu-baseline
b bench_cnn32_parity --model cnn32 --steps 50 --warmup 10 --parity-mode --no-cpu-baseline
b bench_wrn --model wrn --steps 10 --warmup 3 --no-cpu-baseline
b bench_wrn_parity --model wrn --steps 6 --warmup 2 --parity-mode --no-cpu-baseline
b bench_wrn32_bf16 --model wrn --res 32 --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline
b bench_wrn32_parity --model wrn --res 32 --steps 10 --warmup 3 --parity-mode --no-cpu-baseline
timeout -k 10 300 python tools/parity_conv_bench.py 0 > $O/parity_conv_bench.log 2>&1
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
p() { name=$1; shift; timeout -k 10 300 rocprofv3 "$@" > $R/$O/$name.log 2>&1; echo $name rc=$?; }
p stats_wrn --kernel-trace --stats -d $R/$O/stats_wrn -o run -- python $R/bench.py --model wrn --steps 6 --warmup 3 --no-cpu-baseline
p stats_wrn_parity --kernel-trace --stats -d $R/$O/stats_wrn_parity -o run -- python $R/bench.py --model wrn --steps 4 --warmup 2 --no-cpu-baseline --parity-mode
cd $R
db() { find $O/$1 -name "*results.db" 2>/dev/null | head -1; }
for d in stats_wrn stats_wrn_parity; do f=$(db $d); [ -n "$f" ] && python tools/pmc_summary.py stats $f > $O/${d}_kernel_stats.csv; done
rm -rf $O/stats_wrn $O/stats_wrn_parity
ls $O
