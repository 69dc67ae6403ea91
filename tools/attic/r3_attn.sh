cd /root/repo
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -q -m gpu -x -k "attn or attention or block" 2>&1 | tail -3
timeout -k 10 300 python tools/attn_probe.py 2>&1 | grep -v amdgpu.ids | tail -12
