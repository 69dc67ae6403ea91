cd /root/repo
timeout -k 10 900 python -m pytest tests/test_gpu_ops.py -q -m gpu -x -k "gemm_nt or gemm_shapes or epilogue" 2>&1 | tail -2
python tools/nt256_stamps.py 2>&1 | grep -v amdgpu.ids
timeout -k 10 600 python tools/gemm_ab.py 1 129 2>&1 | grep -v amdgpu.ids
