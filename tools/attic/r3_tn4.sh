cd /root/repo
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -q -m gpu -x -k "gemm_tn" 2>&1 | tail -3
echo "== with fetch"; timeout -k 10 200 python tools/tn256_stamps.py 2>&1 | tail -2
echo "== no fetch"; EOE_GEMM_DEBUG=1 timeout -k 10 200 python tools/tn256_stamps.py 2>&1 | tail -2
timeout -k 10 300 python tools/gemm_tn_ab.py 0 4 2>&1 | tail -3
