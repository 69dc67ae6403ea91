# rocprofv3 kernel table of a short serial headline run; prints the rows matching $1 (egrep pattern): bash tools/kstats.sh "multi_reduce|layernorm"
PAT=${1:-.}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/ks && rm -rf /tmp/ks
(cd /tmp && TMPDIR=/tmp timeout -k 10 240 rocprofv3 --kernel-trace --stats -d /tmp/ks -o run -- python $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-torch-baseline --no-roofline --serial-kernels > $R/gpurun_out/ks/rocprof.log 2>&1)
f=$(find /tmp/ks -name "*results.db" | head -1)
if [ -z "$f" ]; then echo "no results.db"; tail -3 $R/gpurun_out/ks/rocprof.log; exit 1; fi
python $R/tools/pmc_summary.py stats $f > $R/gpurun_out/ks/kernel_stats.csv
egrep "$PAT" $R/gpurun_out/ks/kernel_stats.csv | cut -c1-220
