set -o pipefail
O=gpurun_out/r3
mkdir -p $O && cd /root/repo
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -q -m gpu -x -k "gemm_tn" 2>&1 | tail -15 > $O/tn_tests.log; cat $O/tn_tests.log | tail -8
timeout -k 10 300 python tools/gemm_tn_ab.py 0 4 6 > $O/tn_ab.log 2>&1; cat $O/tn_ab.log | tail -8
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-torch-baseline > $O/bench_tn256.json 2> $O/bench_tn256.err && python -c "
import json;d=json.load(open('$O/bench_tn256.json'));print(d['value'],d['ms_per_step'],d['roofline']['kernels_ms_per_step'])"
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-torch-baseline --side-stream 0 > $O/bench_tn256_ss0.json 2> $O/bench_tn256_ss0.err && python -c "
import json;d=json.load(open('$O/bench_tn256_ss0.json'));print(d['value'],d['ms_per_step'],d['roofline']['kernels_ms_per_step'])"
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-torch-baseline --tn-flags 4 > $O/bench_tn128.json 2> $O/bench_tn128.err && python -c "
import json;d=json.load(open('$O/bench_tn128.json'));print(d['value'],d['ms_per_step'],d['roofline']['kernels_ms_per_step'])"
