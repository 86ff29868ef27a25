#!/bin/bash
set -o pipefail
O=$PWD/gpurun_out/${1:-r3h}
mkdir -p $O
export TMPDIR=/tmp
BENCH="python3 $PWD/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras ${2:-}"
ROOT=$PWD
cd /tmp
pass() { timeout -s KILL 120 rocprofv3 --pmc $2 --kernel-trace --output-format csv -d $O/pmc_$1 -- $BENCH > $O/pmc_$1.log 2>&1 || { echo "pmc_$1 failed"; tail -5 $O/pmc_$1.log; exit 1; }; }
pass l2 "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_WRREQ_STALL_sum GRBM_GUI_ACTIVE"
pass tcp1 "TA_TA_BUSY_sum TD_TD_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"
pass tcp2 "TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_WRITEBACK_INVALIDATES_sum"
pass tcc2 "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_WRITE_sum TCC_WRITEBACK_sum"
pass sq "SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_BUSY_CU_CYCLES"
cd $ROOT && python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$O/pmc_*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "k_mix_fir" in r["Kernel_Name"]:
            acc[r["Kernel_Name"][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in acc:
    print(k)
    for n in sorted(acc[k]): print("   ", n, round(sum(acc[k][n]) / len(acc[k][n])))
PY
