# PMC passes over the stand-alone NT GEMM comparison (tools/ntp_check.py --time-only: the shipped kernels against gemm_w8, cold operands):
# L1 -> L2 read latency, L2 hit rate / stalls, SQ wait breakdown, TA busy.  Output: gpurun_out/gemm_pmc/*.txt (mean per launch per kernel)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/gemm_pmc; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
run() { n=$1; shift; rocprofv3 --kernel-trace --pmc "$@" -d $O/$n -o run -- python $R/tools/ntp_check.py fp16 --time-only > $O/$n.log 2>&1; f=$(find $O/$n -name "*_results.db" | head -1); python - "$f" > $O/$n.txt <<'PY'
import sqlite3, sys, collections
con = sqlite3.connect(sys.argv[1])
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for k, c, v, d in con.execute("select kernel_name, counter_name, value, dispatch_id from counters_collection"):
    if "gemm" in k: acc[k][c] += float(v); n[k].add(d)
for k in sorted(acc):
    print(k[:90], "launches", len(n[k]))
    for c, v in sorted(acc[k].items()): print(f"    {c:40s} {v / len(n[k]):16.1f}")
PY
rm -rf $O/$n; }
export NTP_FLAGS=262144
run tcp TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum
run tcc TCC_HIT_sum TCC_MISS_sum TCC_TAG_STALL_sum TCC_REQ_sum
run sq SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES
run ta TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_BUFFER_READ_LDS_WAVEFRONTS_sum
run sq2 SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE
cat $O/*.txt
