#!/bin/bash
# GPU box: HBM traffic counters of the FIR kernel.  Usage: scripts/pmc_traffic.sh <tag>
set -o pipefail
OUT=$PWD/gpurun_out/pmct_$1; mkdir -p $OUT; export TMPDIR=/tmp
BENCH="python3 $PWD/bench.py --steps 3 --warmup 1 --no-cpu-baseline"
cd /tmp
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/a -- $BENCH > $OUT/a.log 2>&1 || { tail -5 $OUT/a.log; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --kernel-trace --output-format csv -d $OUT/b -- $BENCH > $OUT/b.log 2>&1 || { tail -5 $OUT/b.log; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_WRITEBACK_sum --kernel-trace --output-format csv -d $OUT/c -- $BENCH > $OUT/c.log 2>&1 || { tail -5 $OUT/c.log; }
python3 - $OUT <<'PY'
import csv,glob,sys,collections
acc=collections.defaultdict(list)
for f in glob.glob(sys.argv[1]+"/*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "mix_fir" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc): print(k, round(sum(acc[k])/len(acc[k])))
PY
