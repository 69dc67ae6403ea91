# round-5 measurement pass (one gpurun call; ONE generation per round: run it on the final commit and copy gpurun_out/r5m/* to profiles/r5/): GPU suite, PMC passes of the headline command on one stream (the bench line cites
# profiles/r5/bench_pmc_hbm_bytes.csv), every bench line, rocprofv3 kernel tables (the driver's command with its overlap; --serial-kernels = every
# launch on one stream, the table the bench line's roofline object must agree with), per-shape NT GEMM times.  Outputs: gpurun_out/r5m/ (copy the
# summaries to profiles/r5/).  Every step appends to gpurun_out/r5m/progress.log.
O=gpurun_out/r5m
R=$GRAFT_REPO_ROOT
mkdir -p $R/$O && cd $R
say() { echo "$(date +%T) $*" | tee -a $R/$O/progress.log; }
say start
if [ "${SKIP_TESTS:-0}" != 1 ]; then timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; say "pytest: $(tail -1 $O/pytest_gpu.log)"; fi
export TMPDIR=/tmp
db() { find $R/$O/$1 -name "*results.db" | head -1; }
p() { name=$1; shift; (cd /tmp && timeout -k 10 300 rocprofv3 "$@" > $R/$O/$name.log 2>&1); say "$name rc=$?"; }
HL="--steps 3 --warmup 2 --no-cpu-baseline --no-torch-baseline --no-roofline --no-box-probe --serial-kernels"      # (no probe launches in a profiled run)
p pmc_fetch --kernel-trace --pmc FETCH_SIZE -d $R/$O/pmc_fetch -o run -- python $R/bench.py $HL
p pmc_write --kernel-trace --pmc WRITE_SIZE -d $R/$O/pmc_write -o run -- python $R/bench.py $HL
p pmc_sq --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $R/$O/pmc_sq -o run -- python $R/bench.py $HL
p pmc_tcp --kernel-trace --pmc TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCC_HIT_sum TCC_MISS_sum -d $R/$O/pmc_tcp -o run -- python $R/bench.py $HL
ff=$(db pmc_fetch); fw=$(db pmc_write); fs=$(db pmc_sq); ft=$(db pmc_tcp)
mkdir -p profiles/r5
python tools/pmc_summary.py hbm $ff $fw > $O/bench_pmc_hbm_bytes.csv && cp $O/bench_pmc_hbm_bytes.csv profiles/r5/bench_pmc_hbm_bytes.csv      # (bench.py reads the newest round's)
python tools/pmc_summary.py sq $fs > $O/bench_pmc_sq.csv
python tools/pmc_summary.py sq $ft > $O/bench_pmc_tcp.csv
rm -rf $O/pmc_fetch $O/pmc_write $O/pmc_sq $O/pmc_tcp
b() { name=$1; shift; timeout -k 10 300 python bench.py "$@" > $O/$name.json 2> $O/$name.err; say "$name $(cut -c1-150 $O/$name.json)"; }
b bench --steps 20 --warmup 5
b bench_serial --steps 20 --warmup 5 --serial-kernels --no-cpu-baseline --no-torch-baseline
b bench_frozen --mode frozen --no-cpu-baseline --no-torch-baseline
b bench_eval --mode eval --no-cpu-baseline
b bench_bf16 --dtype bf16 --no-cpu-baseline --no-torch-baseline
b bench_cnn32 --model cnn32 --steps 50 --warmup 10 --no-cpu-baseline
b bench_cnn32_parity --model cnn32 --steps 50 --warmup 10 --no-cpu-baseline --parity-mode
b bench_wrn --model wrn --steps 10 --warmup 3 --no-cpu-baseline
b bench_wrn_parity --model wrn --steps 10 --warmup 3 --no-cpu-baseline --parity-mode
b bench_wrn32_bf16 --model wrn --res 32 --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline
b bench_wrn32_parity --model wrn --res 32 --steps 20 --warmup 5 --no-cpu-baseline --parity-mode
p stats_default --kernel-trace --stats -d $R/$O/stats_default -o run -- python $R/bench.py --steps 8 --warmup 5 --no-cpu-baseline --no-torch-baseline --no-roofline --no-box-probe
p stats --kernel-trace --stats -d $R/$O/stats -o run -- python $R/bench.py --steps 8 --warmup 5 --no-cpu-baseline --no-torch-baseline --no-roofline --no-box-probe --serial-kernels
for d in stats_default stats; do f=$(db $d); [ -n "$f" ] && python tools/pmc_summary.py stats $f > $O/${d}_kernel_stats.csv; done
rm -rf $O/stats_default $O/stats
EOE_PROF_SHAPES=1 timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-torch-baseline --no-box-probe 2>/dev/null > $O/shapes.json
python - <<'PY' > gpurun_out/r5m/nt_shapes.txt
import json
d = json.loads(open("gpurun_out/r5m/shapes.json").read())
k = d["roofline"]["kernels_ms_per_step"]
print("per-shape NT GEMM times inside the step's one-stream pass (ms per step over all launches of the shape; e0 plain, e1 GELU pair, e2 fp32 residual, e3 GELU' x dY, c = fused column sums)")
for n, v in sorted(k.items(), key=lambda kv: -kv[1]):
    if n.startswith("nt_"): print(f"{n:36s} {v:7.3f}")
PY
timeout -k 10 120 python bench.py --batch 2 --steps 60 --warmup 10 --no-cpu-baseline --no-torch-baseline --no-roofline --no-box-probe > $O/bench_host_only.json 2>/dev/null; say "host only $(cut -c1-200 $O/bench_host_only.json)"
timeout -k 10 200 python tools/host_time.py 40 > $O/host_time.log 2>&1
say "done: $(head -3 $O/stats_kernel_stats.csv | cut -c1-120)"
