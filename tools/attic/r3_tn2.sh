set -o pipefail
O=gpurun_out/r3
mkdir -p $O && cd /root/repo
echo "== no-fetch (EOE_GEMM_DEBUG=1)"; EOE_GEMM_DEBUG=1 timeout -k 10 300 python tools/gemm_tn_ab.py 0 4 2>&1 | tail -3
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for v in 0 4; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/$O/tnpmc_f$v -o run -- python $R/tools/gemm_tn_ab.py $v > $R/$O/tnpmc_f$v.log 2>&1; echo fetch $v rc=$?
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/$O/tnpmc_w$v -o run -- python $R/tools/gemm_tn_ab.py $v > $R/$O/tnpmc_w$v.log 2>&1; echo write $v rc=$?
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $R/$O/tnpmc_s$v -o run -- python $R/tools/gemm_tn_ab.py $v > $R/$O/tnpmc_s$v.log 2>&1; echo sq $v rc=$?
done
cd $R
for v in 0 4; do
  ff=$(find $O/tnpmc_f$v -name "*results.db" | head -1); fw=$(find $O/tnpmc_w$v -name "*results.db" | head -1); fs=$(find $O/tnpmc_s$v -name "*results.db" | head -1)
  echo "== variant $v"; python tools/pmc_summary.py hbm $ff $fw | grep -i "gemm_tn" | head -4; python tools/pmc_summary.py sq $fs | grep -i "gemm_tn\|kernel" | head -4
done
rm -rf $O/tnpmc_*
