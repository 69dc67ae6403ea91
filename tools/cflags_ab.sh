# the library built with and without extra compiler flags, interleaved bench runs on one box: bash tools/cflags_ab.sh "-mllvm -amdgpu-kernarg-preload-count=16"
F="$1"
ROOF=${ROOF---no-roofline}     # ROOF="" keeps the per-kernel pass: the summary then lists kernels_ms_per_step too
R=$GRAFT_REPO_ROOT; cd $R
mkdir -p gpurun_out/cf /tmp/cf
cp eoe_amd/libeoe_hip.so /tmp/cf/base.so; cp eoe_amd/libeoe_hip.so.stamp /tmp/cf/base.stamp
EOE_CFLAGS="$F" python -c "from eoe_amd import _build; _build.build(force=True, verbose=False)" > gpurun_out/cf/build.log 2>&1 || { tail -5 gpurun_out/cf/build.log; exit 1; }
cp eoe_amd/libeoe_hip.so /tmp/cf/flag.so; cp eoe_amd/libeoe_hip.so.stamp /tmp/cf/flag.stamp
echo "built with: $F"
for r in 1 2 3; do
cp /tmp/cf/base.so eoe_amd/libeoe_hip.so; cp /tmp/cf/base.stamp eoe_amd/libeoe_hip.so.stamp
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-torch-baseline $ROOF 2>/dev/null > gpurun_out/cf/base_$r.json
cp /tmp/cf/flag.so eoe_amd/libeoe_hip.so; cp /tmp/cf/flag.stamp eoe_amd/libeoe_hip.so.stamp
EOE_CFLAGS="$F" python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-torch-baseline $ROOF 2>/dev/null > gpurun_out/cf/flag_$r.json
done
python - <<'PY'
import json
for t in ("base", "flag"):
    v = [json.loads(open(f"gpurun_out/cf/{t}_{r}.json").read())["ms_per_step"] for r in (1, 2, 3)]
    print(t, v, sum(v) / 3)
    ks = [(json.loads(open(f"gpurun_out/cf/{t}_{r}.json").read()).get("roofline") or {}).get("kernels_ms_per_step") for r in (1, 2, 3)]
    if all(ks):
        print("   ", {k: round(sum(x[k] for x in ks) / 3, 3) for k in ks[0] if ks[0][k] >= 0.05})
PY
