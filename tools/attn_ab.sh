set -e
cd $GRAFT_REPO_ROOT
# old (tools/ab/prev) vs new build: per-kernel in-step times from the bench's serialised roofline pass, interleaved
bash tools/ab/run_ab.sh --steps 60 --warmup 10 --no-cpu-baseline --no-torch-baseline > /dev/null
python - <<'PY'
import json
for v in ("old", "new"):
    for r in (1, 2, 3):
        d = json.loads(open(f"gpurun_out/sh/ab_{v}_{r}.json").read())
        k = d["roofline"]["kernels_ms_per_step"]
        print(v, r, d["ms_per_step"], "attn_bwd", k["attn_bwd"], "gemm_nt", k["gemm_nt"])
PY
