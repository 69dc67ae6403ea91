# round-3 check pass: GPU tests + the default bench line + eval line (run through gpurun from the repo root)
set -o pipefail
O=gpurun_out/r3
mkdir -p $O && cd /root/repo
timeout -k 10 1100 python -m pytest tests -q -m gpu -x 2>&1 | tail -40 > $O/tests.log; cat $O/tests.log | tail -15
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err && cut -c1-400 $O/bench.json
timeout -k 10 200 python bench.py --mode eval --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_eval.json 2> $O/bench_eval.err && cut -c1-300 $O/bench_eval.json
