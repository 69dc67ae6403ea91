set -e
mkdir -p gpurun_out/ntw
for r in 1 2 3; do for v in 0 4096; do
  timeout -k 10 200 python bench.py --steps 200 --warmup 20 --nt-flags $v --no-roofline --no-cpu-baseline --no-torch-baseline > gpurun_out/ntw/b_${v}_${r}.json 2> gpurun_out/ntw/b_${v}_${r}.err
  python -c "
import json
d=json.loads(open('gpurun_out/ntw/b_${v}_${r}.json').read().strip().splitlines()[-1]); print('nt_flags=$v run $r: %.3f ms %.0f img/s loss %s' % (d['ms_per_step'], d['value'], d['final_loss']), flush=True)"
done; done
