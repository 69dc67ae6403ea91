#!/bin/bash
# A/B of the last block on its class-token rows only (EOE_VIT_CLS_ONLY) on one box: alternating bench runs, timed pass only
set -e
mkdir -p gpurun_out/cls
for r in 1 2 3; do
  for v in 0 1; do
    EOE_VIT_CLS_ONLY=$v timeout -k 10 200 python bench.py --steps 300 --warmup 30 --no-roofline --no-cpu-baseline > gpurun_out/cls/b_${v}_${r}.json 2> gpurun_out/cls/b_${v}_${r}.err
    python - <<PY
import json
d=json.loads(open("gpurun_out/cls/b_${v}_${r}.json").read().strip().splitlines()[-1])
print("cls_only=${v} run ${r}: %.3f ms  %.0f img/s" % (d["ms_per_step"], d["value"]), flush=True)
PY
  done
done
