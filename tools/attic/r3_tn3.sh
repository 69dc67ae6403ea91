cd /root/repo
echo "== with fetch"; timeout -k 10 200 python tools/tn256_stamps.py 2>&1 | tail -2
echo "== no fetch"; EOE_GEMM_DEBUG=1 timeout -k 10 200 python tools/tn256_stamps.py 2>&1 | tail -2
