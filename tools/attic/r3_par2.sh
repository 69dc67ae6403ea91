cd /root/repo
O=gpurun_out/r3; mkdir -p $O
for cfg in "wrn --steps 6 --warmup 2" "cnn32 --steps 30 --warmup 5" "wrn --res 32 --steps 10 --warmup 3"; do
  n=$(echo $cfg | tr ' -' '__')
  timeout -k 10 300 python bench.py --model $cfg --parity-mode --no-cpu-baseline > $O/par_$n.json 2> $O/par_$n.err && python -c "
import json;d=json.load(open('$O/par_$n.json'));r=d['roofline'];print('$cfg', d['value'],d['ms_per_step'],r['kernel'],r['achieved'],r['frac']);print({k:v for k,v in sorted(r['kernels_ms_per_step'].items(), key=lambda x:-x[1])[:8]})" || tail -3 $O/par_$n.err
done
timeout -k 10 900 python -m pytest tests/test_gpu_parity_big.py tests/test_gpu_resnet.py tests/test_gpu_cnn.py -q -m gpu -x -k "parity" 2>&1 | tail -4
