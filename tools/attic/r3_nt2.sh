cd /root/repo
timeout -k 10 900 python -m pytest tests/test_gpu_ops.py -q -m gpu -x -k "gemm_nt_plain and one_wave_160" 2>&1 | grep -v "^$" | tail -30
