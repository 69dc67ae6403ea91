set -o pipefail
O=gpurun_out/tl; mkdir -p $O; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace -d $R/$O/tr -o run -- python $R/bench.py --steps 8 --warmup 4 --no-cpu-baseline --no-roofline --no-torch-baseline > $R/$O/tr.log 2>&1
cd $R; f=$(find $O/tr -name "*results.db" | head -1); python tools/timeline.py $f > $O/timeline.txt; rm -rf $O/tr; cat $O/timeline.txt
