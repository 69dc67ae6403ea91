# interleaved A/B of one bench.py option on ONE box: bash tools/ab_opt2.sh "--vit-flags 1" "--vit-flags 0" [pairs]
A="$1"; B="$2"; N=${3:-3}
mkdir -p gpurun_out/ab2
for r in $(seq 1 $N); do
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-torch-baseline --no-roofline $A 2>/dev/null > gpurun_out/ab2/a_$r.json
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-torch-baseline --no-roofline $B 2>/dev/null > gpurun_out/ab2/b_$r.json
done
python - "$A" "$B" "$N" <<'PY'
import json, sys
A, B, N = sys.argv[1], sys.argv[2], int(sys.argv[3])
for tag, name in (("a", A), ("b", B)):
    v = [json.loads(open(f"gpurun_out/ab2/{tag}_{r}.json").read())["ms_per_step"] for r in range(1, N + 1)]
    print(f"{name:>24s}: ms per step {v}  mean {sum(v)/len(v):.3f}")
PY
