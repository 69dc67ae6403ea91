# (the per-kernel profiles are taken with --serial-kernels: overlapped launches stretch each other's durations; stats_default = the
#  driver's exact command, with the overlap)
# round-3 final measurement pass: every bench line + rocprofv3 kernel tables + PMC passes of the headline command, after the late-round
# changes (per-shape NT kernel choice, backward hand-over, asynchronous weight gradients).  Outputs under gpurun_out/r3h/.
set -o pipefail
O=gpurun_out/r3h
mkdir -p $O && cd /root/repo
b() { name=$1; shift; timeout -k 10 300 python bench.py "$@" > $O/$name.json 2> $O/$name.err && cut -c1-160 $O/$name.json; }
b bench --steps 20 --warmup 5
b bench_frozen --mode frozen --no-cpu-baseline --no-torch-baseline
b bench_eval --mode eval --no-cpu-baseline
b bench_bf16 --dtype bf16 --no-cpu-baseline --no-torch-baseline
b bench_cnn32 --model cnn32 --steps 50 --warmup 10 --no-cpu-baseline
b bench_cnn32_parity --model cnn32 --steps 50 --warmup 10 --parity-mode --no-cpu-baseline
b bench_wrn --model wrn --steps 10 --warmup 3 --no-cpu-baseline
b bench_wrn_parity --model wrn --steps 6 --warmup 2 --parity-mode --no-cpu-baseline
b bench_wrn32_bf16 --model wrn --res 32 --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline
b bench_wrn32_parity --model wrn --res 32 --steps 10 --warmup 3 --parity-mode --no-cpu-baseline
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
p() { name=$1; shift; timeout -k 10 300 rocprofv3 "$@" > $R/$O/$name.log 2>&1; echo $name rc=$?; }
p stats_default --kernel-trace --stats -d $R/$O/stats_default -o run -- python $R/bench.py --steps 8 --warmup 5 --no-cpu-baseline --no-torch-baseline
p stats --kernel-trace --stats -d $R/$O/stats -o run -- python $R/bench.py --steps 8 --warmup 5 --no-cpu-baseline --no-torch-baseline --serial-kernels
p stats_wrn --kernel-trace --stats -d $R/$O/stats_wrn -o run -- python $R/bench.py --model wrn --steps 6 --warmup 3 --no-cpu-baseline --serial-kernels
p stats_wrn_parity --kernel-trace --stats -d $R/$O/stats_wrn_parity -o run -- python $R/bench.py --model wrn --steps 4 --warmup 2 --no-cpu-baseline --parity-mode --serial-kernels
p pmc_fetch --kernel-trace --pmc FETCH_SIZE -d $R/$O/pmc_fetch -o run -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-torch-baseline --no-roofline --serial-kernels
p pmc_write --kernel-trace --pmc WRITE_SIZE -d $R/$O/pmc_write -o run -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-torch-baseline --no-roofline --serial-kernels
p pmc_sq --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $R/$O/pmc_sq -o run -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-torch-baseline --no-roofline --serial-kernels
cd $R
db() { find $O/$1 -name "*results.db" 2>/dev/null | head -1; }
for d in stats_default stats stats_wrn stats_wrn_parity; do f=$(db $d); [ -n "$f" ] && python tools/pmc_summary.py stats $f > $O/${d}_kernel_stats.csv; done
ff=$(db pmc_fetch); fw=$(db pmc_write); fs=$(db pmc_sq)
python tools/pmc_summary.py hbm $ff $fw > $O/bench_pmc_hbm_bytes.csv
python tools/pmc_summary.py sq $fs > $O/bench_pmc_sq.csv
rm -rf $O/stats_default $O/stats $O/stats_wrn $O/stats_wrn_parity $O/pmc_fetch $O/pmc_write $O/pmc_sq
timeout -k 10 600 python -m pytest tests/test_gpu_parity_big.py -q -s > $O/parity_big.log 2>&1; echo parity rc=$?
ls $O | head -40
