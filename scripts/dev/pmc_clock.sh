#!/bin/bash
# GPU box: average shader clock during the FIR kernel = GRBM_GUI_ACTIVE / 8 XCDs / kernel duration.
# Usage: scripts/pmc_clock.sh <tag> [library]   (library: an ablation build of libatsc_hip.so, default the product library)
set -o pipefail
OUT=$PWD/gpurun_out/clk_$1; mkdir -p $OUT; export TMPDIR=/tmp
[ -n "$2" ] && export ATSC_HIP_LIB=$PWD/$2
BENCH="python3 $PWD/bench.py --steps 5 --warmup 2 --no-cpu-baseline"
cd /tmp
timeout -s KILL 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $OUT/a -- $BENCH > $OUT/a.log 2>&1 || { tail -5 $OUT/a.log; exit 1; }
python3 - $OUT <<'PY'
import csv,glob,sys,collections
acc=collections.defaultdict(list); dur=[]
for f in glob.glob(sys.argv[1]+"/*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "mix_fir" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob(sys.argv[1]+"/*/*/*_kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        if "mix_fir" in r["Kernel_Name"]: dur.append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
for k in sorted(acc): print(k, round(sum(acc[k])/len(acc[k])))
if dur:
    d=sum(dur)/len(dur); g=sum(acc["GRBM_GUI_ACTIVE"])/len(acc["GRBM_GUI_ACTIVE"])
    print("avg_us", round(d,1), "clock_GHz", round(g/8/d/1e3,3))
PY
