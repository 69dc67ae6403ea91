cd /root/repo
timeout -k 10 600 python -m pytest tests/test_gpu_parity_big.py -q -m gpu -x -s -k "wideresnet32_fast" 2>&1 | grep "loss dev\|ref noise\|score dev\|allowed\|Assert\|passed\|failed" | cut -c1-420
bash tools/r3_attn.sh
