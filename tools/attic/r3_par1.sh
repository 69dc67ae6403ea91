cd /root/repo
timeout -k 10 900 python -m pytest tests/test_gpu_parity_big.py -q -m gpu -x -k "conv_f32_kernels" 2>&1 | tail -15
