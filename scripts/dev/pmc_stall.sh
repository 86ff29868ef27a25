#!/bin/bash
# GPU box: where do the FIR kernel's waves wait?  Store-path / LDS FIFO-full counters (SQ), address/data stalls (TA, TCP,
# L1 TLB) and write-request stalls at the L2's memory side (TCC).  Usage: scripts/pmc_stall.sh <tag> [library]
set -o pipefail
OUT=$PWD/gpurun_out/stall_$1; mkdir -p $OUT; export TMPDIR=/tmp
[ -n "$2" ] && export ATSC_HIP_LIB=$PWD/$2
BENCH="python3 $PWD/bench.py --steps 3 --warmup 1 --no-cpu-baseline"
cd /tmp
run() { timeout -s KILL 120 rocprofv3 --pmc $2 --kernel-trace --output-format csv -d $OUT/$1 -- $BENCH > $OUT/$1.log 2>&1 || { tail -5 $OUT/$1.log; exit 1; }; }
run a "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_WAIT_INST_LDS SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES"
run b "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_SERIALIZATION_STALL_sum"
run c "TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_BUSY_sum"
run d "TA_TA_BUSY_sum TD_TD_BUSY_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCP_TA_ADDR_STALL_CYCLES_sum TCP_WRITE_TAGCONFLICT_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum GRBM_GUI_ACTIVE"
python3 - $OUT <<'PY'
import csv,glob,sys,collections
acc=collections.defaultdict(list)
for f in glob.glob(sys.argv[1]+"/*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "mix_fir" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc): print(k, round(sum(acc[k])/len(acc[k])))
PY
