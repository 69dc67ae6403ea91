# (round 5) could a split-operand 16-bit MFMA product replace the exact fp32 matrix-core convolutions of parity mode?
# EOE_PARITY_EMULATE_BITS=b rounds both operands of every parity-mode forward convolution to b explicit mantissa bits (15: bf16 hi + lo, 21: fp16
# hi + lo) before the exact fp32 product: a LOWER bound on the noise of hi.hi + hi.lo + lo.hi.  The two fixtures at the stated bar, unchanged.
mkdir -p gpurun_out
for b in 0 21 15; do
  echo "=== EOE_PARITY_EMULATE_BITS=$b"
  EOE_PARITY_EMULATE_BITS=$b timeout -k 10 500 python -m pytest tests/test_gpu_parity_big.py -q -s -m gpu -k "test_cnn32_big_parity_mode or (test_wideresnet_full_batch and parity)" 2>&1 | grep -E "PARITY|loss dev|ref noise|score dev|allowed|passed|failed|AssertionError|worst" | cut -c1-330
done
