set -o pipefail
# the fp16-conv-output speed option (ops.CONV_Y16, bench.py --conv-y16): tests that cover it + bench lines with and without
O=gpurun_out/y16; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_resnet.py tests/test_gpu_cnn.py -q -m gpu > $O/pytest.log 2>&1; tail -2 $O/pytest.log | cut -c1-300
timeout -k 10 600 python -m pytest tests/test_gpu_parity_big.py -q -s -m gpu -k "wideresnet_full_batch" > $O/parity_full.log 2>&1; tail -2 $O/parity_full.log
for v in "" "--conv-y16"; do
  timeout -k 10 250 python bench.py --model wrn --steps 30 --warmup 5 --no-cpu-baseline --no-torch-baseline $v > $O/wrn$v.json 2> $O/wrn$v.err && cut -c1-170 $O/wrn$v.json
  timeout -k 10 250 python bench.py --model cnn32 --steps 200 --warmup 20 --no-cpu-baseline --no-torch-baseline $v > $O/cnn32$v.json 2> $O/cnn32$v.err && cut -c1-170 $O/cnn32$v.json
done
