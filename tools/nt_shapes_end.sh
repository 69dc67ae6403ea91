# per-shape in-step times of the NT GEMMs (serialised roofline pass of bench.py, EOE_PROF_SHAPES=1) for the final build: default kernel choice,
# with the 160x256x32 two-workgroup kernel off (nt_flags 4096), and with every shape on the one-wave kernel (nt_flags 512); interleaved twice
mkdir -p gpurun_out/sh
for r in 1 2; do for f in 0 4096 512; do
EOE_PROF_SHAPES=1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-torch-baseline --nt-flags $f 2>/dev/null > gpurun_out/sh/e_${f}_$r.json
done; done
python - <<'PY'
import json
res = {}
for f in (0, 4096, 512):
    for r in (1, 2):
        d = json.loads(open(f"gpurun_out/sh/e_{f}_{r}.json").read())
        k = d["roofline"]["kernels_ms_per_step"]
        for n, v in k.items():
            if n.startswith("nt_"):
                res.setdefault(n, {}).setdefault(f, []).append(v)
        print(f"nt_flags {f} run {r}: {d['ms_per_step']} ms per step")
print("shape (M x N x K _epilogue: 0 plain, 1 GELU pair, 2 fp32 residual, 3 GELU' x dY; c = fused column sums): ms per step over all launches of the shape")
av = lambda x: sum(x) / max(1, len(x))
for n in sorted(res, key=lambda n: -av(res[n].get(0, [0]))):
    print(f"{n:34s} default {av(res[n].get(0, [0])):7.3f}   without the wide two-workgroup kernel {av(res[n].get(4096, [0])):7.3f}   all one-wave {av(res[n].get(512, [0])):7.3f}")
PY
