cd /root/repo
EOE_PARITY_REPORT=1 timeout -k 10 600 python -m pytest tests/test_gpu_parity_big.py -q -m gpu -x -s -k "frozen_ranking" 2>&1 | grep "frozen ranking\|AUC of\|passed\|failed" | cut -c1-900
