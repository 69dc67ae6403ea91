# per-shape in-step times of the NT GEMMs under the two-workgroup kernel (nt_flags 0) and the one-wave kernel (nt_flags 512), interleaved
mkdir -p gpurun_out/sh
for r in 1 2; do for f in 0 512; do
EOE_PROF_SHAPES=1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-torch-baseline --nt-flags $f 2>/dev/null > gpurun_out/sh/s_${f}_$r.json
done; done
python - <<'PY'
import json, glob
res = {}
for f in (0, 512):
    for r in (1, 2):
        d = json.loads(open(f"gpurun_out/sh/s_{f}_{r}.json").read())
        k = d["roofline"]["kernels_ms_per_step"]
        for n, v in k.items():
            if n.startswith("nt_"):
                res.setdefault(n, {}).setdefault(f, []).append(v)
        print(f, r, d["ms_per_step"])
for n in sorted(res, key=lambda n: -sum(res[n].get(0, [0]))):
    a, b = res[n].get(0, [0]), res[n].get(512, [0])
    print(f"{n:34s} two-wg {sum(a)/len(a):7.3f}   one-wave {sum(b)/len(b):7.3f}   diff {sum(b)/len(b)-sum(a)/len(a):+.3f} ms/step")
PY
