# in-step attention kernel times against an occupancy limit (extra dynamic LDS per workgroup): bash tools/attn_pad.sh
mkdir -p gpurun_out/ap
for r in 1 2; do for pf in 0 8192 16384 32768; do
EOE_ATTN_PAD_FWD=$pf EOE_ATTN_PAD_BWD=$pf python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-torch-baseline 2>/dev/null > gpurun_out/ap/p${pf}_$r.json
python - $pf $r <<'PY'
import json, sys
d = json.loads(open(f"gpurun_out/ap/p{sys.argv[1]}_{sys.argv[2]}.json").read())
k = d["roofline"]["kernels_ms_per_step"]
print("pad", sys.argv[1], "ms", d["ms_per_step"], "attn_fwd", k.get("attn_fwd"), "attn_bwd", k.get("attn_bwd"), flush=True)
PY
done; done
