cd /root/repo
O=gpurun_out/r3
mkdir -p $O
for f in "" "--side-stream 0" "--tn-flags 4" "--side-stream 2"; do
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-torch-baseline $f > $O/b.json 2> $O/b.err && python -c "
import json;d=json.load(open('$O/b.json'));k=d['roofline']['kernels_ms_per_step'];print('$f', d['value'],d['ms_per_step'],'tn',k['gemm_tn'],'nt',k['gemm_nt'],'lnb',k['layernorm_bwd'])"
done
