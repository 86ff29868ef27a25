#!/bin/bash
# GPU box: SQ counters for the FIR kernel (MFMA form).  Usage: scripts/pmc_mfma.sh <tag>
set -o pipefail
OUT=$PWD/gpurun_out/pmc_$1; mkdir -p $OUT; export TMPDIR=/tmp
BENCH="python3 $PWD/bench.py --steps 3 --warmup 1 --no-cpu-baseline"
cd /tmp
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/a -- $BENCH > $OUT/a.log 2>&1 || { tail -5 $OUT/a.log; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d $OUT/b -- $BENCH > $OUT/b.log 2>&1 || { tail -5 $OUT/b.log; exit 1; }
python3 - $OUT <<'PY'
import csv,glob,sys,collections
acc=collections.defaultdict(list)
for f in glob.glob(sys.argv[1]+"/*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "mix_fir" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc): print(k, round(sum(acc[k])/len(acc[k])))
PY
