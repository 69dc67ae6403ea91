cd /root/repo
timeout -k 10 1100 python -m pytest tests/test_gpu_parity_big.py -q -m gpu -x -s -k "frozen or cnn32_big or cnn28_big or wideresnet_big or wideresnet32 or wideresnet_full" 2>&1 | grep -v "^$" | grep "frozen ranking\|passed\|failed\|Error\|assert\|allowed" | cut -c1-400 | tail -40
