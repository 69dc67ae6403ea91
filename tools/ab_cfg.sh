# old (tools/ab/prev) vs new build on several bench configurations, one box: bash tools/ab_cfg.sh
set -e
for cfg in "--model wrn --steps 10 --warmup 3 --no-cpu-baseline --parity-mode" "--model wrn --steps 10 --warmup 3 --no-cpu-baseline" "--model wrn --res 32 --steps 20 --warmup 5 --no-cpu-baseline --parity-mode" "--model cnn32 --steps 50 --warmup 10 --no-cpu-baseline --parity-mode"; do
echo "== $cfg"
bash tools/ab/run_ab.sh $cfg --no-torch-baseline 2>&1 | head -12
done
