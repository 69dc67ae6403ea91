# forward-only scoring: step time against the sum of its kernels' durations (rocprofv3 kernel trace): how much is launch / host, not kernels
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/eg; rm -rf /tmp/eg
python bench.py --mode eval --steps 30 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | cut -c1-200
(cd /tmp && TMPDIR=/tmp timeout -k 10 240 rocprofv3 --kernel-trace --stats -d /tmp/eg -o run -- python $R/bench.py --mode eval --steps 30 --warmup 5 --no-cpu-baseline --no-roofline > $R/gpurun_out/eg/rocprof.log 2>&1)
f=$(find /tmp/eg -name "*results.db" | head -1)
python tools/pmc_summary.py stats $f > gpurun_out/eg/kernel_stats.csv
python - <<'PY'
import csv
rows = list(csv.DictReader(open("gpurun_out/eg/kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows); calls = sum(int(r["Calls"]) for r in rows)
print(f"sum of kernel durations per step (35 steps): {tot / 35e6:.3f} ms, {calls / 35:.0f} launches per step")
for r in rows[:8]: print(r["Name"][:60], r["Calls"], f"{float(r['AverageNs']) / 1e3:.1f} us")
PY
