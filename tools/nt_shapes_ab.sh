# per-shape in-step NT GEMM times (serialised roofline pass, EOE_PROF_SHAPES=1) for two nt_flags settings, interleaved: bash tools/nt_shapes_ab.sh A B
A=${1:-524288}; B=${2:-0}
mkdir -p gpurun_out/sh
for r in 1 2; do for f in $A $B; do
EOE_PROF_SHAPES=1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-torch-baseline --nt-flags $f 2>/dev/null > gpurun_out/sh/s_${f}_$r.json
done; done
python - $A $B <<'PY'
import json, sys
A, B = sys.argv[1], sys.argv[2]
res = {}
for f in (A, B):
    for r in (1, 2):
        d = json.loads(open(f"gpurun_out/sh/s_{f}_{r}.json").read())
        for n, v in d["roofline"]["kernels_ms_per_step"].items():
            if n.startswith("nt_"):
                res.setdefault(n, {}).setdefault(f, []).append(v)
        print(f"nt_flags {f} run {r}: {d['ms_per_step']} ms per step")
av = lambda x: sum(x) / max(1, len(x))
for n in sorted(res, key=lambda n: -av(res[n].get(A, [0])))[:10]:
    print(f"{n:32s} {A}: {av(res[n].get(A, [0])):.3f}   {B}: {av(res[n].get(B, [0])):.3f}")
PY
