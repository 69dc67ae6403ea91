set -o pipefail
O=gpurun_out/pw; mkdir -p $O
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/$O/st -o run -- python $R/bench.py --model wrn --steps 6 --warmup 3 --no-cpu-baseline --no-roofline > $R/$O/log 2>&1
cd $R; f=$(find $O/st -name "*results.db" | head -1); python tools/pmc_summary.py stats $f > $O/wrn_stats.csv; rm -rf $O/st; head -45 $O/wrn_stats.csv | cut -c1-150
