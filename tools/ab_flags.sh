# interleaved A/B of two nt_flags settings on ONE box: bash tools/ab_flags.sh FLAGS_A FLAGS_B [pairs] [extra bench.py arguments]
A=${1:-131072}; B=${2:-0}; N=${3:-3}; shift 3 2>/dev/null
mkdir -p gpurun_out/ab
for r in $(seq 1 $N); do for f in $A $B; do
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-torch-baseline --no-roofline --nt-flags $f "$@" 2>/dev/null > gpurun_out/ab/f${f}_$r.json
done; done
python - "$A" "$B" "$N" <<'PY'
import json, sys
A, B, N = sys.argv[1], sys.argv[2], int(sys.argv[3])
for f in (A, B):
    v = [json.loads(open(f"gpurun_out/ab/f{f}_{r}.json").read())["ms_per_step"] for r in range(1, N + 1)]
    print(f"nt_flags {f:>7s}: ms per step {v}  mean {sum(v)/len(v):.3f}")
PY
