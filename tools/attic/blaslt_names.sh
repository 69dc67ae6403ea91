set -o pipefail
O=gpurun_out/blt; mkdir -p $O; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $R/$O/tr -o run -- python $R/tools/blaslt_names.py > $R/$O/tr.log 2>&1
cd $R; f=$(find $O/tr -name "*results.db" | head -1); python - "$f" <<'PY'
import sqlite3, sys
con = sqlite3.connect(sys.argv[1])
rows = list(con.execute("select name, count(*), avg(duration), min(duration), grid_x, grid_y, workgroup_x, lds_size, vgpr_count, accum_vgpr_count from kernels group by name order by avg(duration) desc"))
for r in rows:
    if "Cijk" in r[0] or "gemm" in r[0].lower():
        print(r[1], "calls avg %.1f us min %.1f us grid %s x %s wg %s lds %s vgpr %s agpr %s" % (r[2] / 1e3, r[3] / 1e3, r[4], r[5], r[6], r[7], r[8], r[9]))
        print("   ", r[0][:400])
PY
rm -rf $O/tr
