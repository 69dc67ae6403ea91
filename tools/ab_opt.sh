# interleaved A/B of two values of one bench.py option on ONE box: bash tools/ab_opt.sh --tn-flags 8 0 [pairs] [extra bench.py arguments]
OPT=$1; A=$2; B=$3; N=${4:-3}; shift 4 2>/dev/null
mkdir -p gpurun_out/ab
for r in $(seq 1 $N); do for f in $A $B; do
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-torch-baseline --no-roofline $OPT $f "$@" 2>/dev/null > gpurun_out/ab/o${f}_$r.json
done; done
python - "$OPT" "$A" "$B" "$N" <<'PY'
import json, sys
OPT, A, B, N = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
for f in (A, B):
    v = [json.loads(open(f"gpurun_out/ab/o{f}_{r}.json").read())["ms_per_step"] for r in range(1, N + 1)]
    print(f"{OPT} {f:>7s}: ms per step {v}  mean {sum(v)/len(v):.3f}")
PY
