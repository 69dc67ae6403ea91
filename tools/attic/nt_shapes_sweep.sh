# per-shape in-step times of the NT GEMMs under each tile-shape variant (nt_flags), two interleaved rounds on one box
mkdir -p gpurun_out/sh
V="${1:-0 4 16 32 128 256}"
for r in 1 2; do for f in $V; do
EOE_PROF_SHAPES=1 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-torch-baseline --nt-flags $f 2>/dev/null > gpurun_out/sh/w_${f}_$r.json
done; done
V="$V" python - <<'PY'
import json, os
V = [int(v) for v in os.environ["V"].split()]
res, steps = {}, {}
for f in V:
    for r in (1, 2):
        d = json.loads(open(f"gpurun_out/sh/w_{f}_{r}.json").read())
        steps.setdefault(f, []).append(d["ms_per_step"])
        for n, v in d["roofline"]["kernels_ms_per_step"].items():
            if n.startswith("nt_128") or n.startswith("nt_125"):
                res.setdefault(n, {}).setdefault(f, []).append(v)
print("step ms:", {f: [round(x, 3) for x in steps[f]] for f in V})
print(f"{'shape':30s}" + "".join(f"{f:>9d}" for f in V))
for n in sorted(res, key=lambda n: -sum(res[n].get(V[0], [0]))):
    print(f"{n:30s}" + "".join(f"{sum(res[n].get(f,[0]))/max(1,len(res[n].get(f,[0]))):9.3f}" for f in V))
PY
