#!/bin/bash
# Run ON THE GPU BOX: kernel-trace stats only.  Usage: scripts/trace_gpu.sh <tag>
set -o pipefail
TAG=${1:-t}
OUT=$PWD/gpurun_out/trace_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python3 $PWD/bench.py --steps 5 --warmup 2 --no-cpu-baseline"
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- $BENCH > $OUT/trace.log 2>&1 || { echo "trace failed"; tail -5 $OUT/trace.log; exit 1; }
cat $OUT/*/*_kernel_stats.csv | cut -d, -f1-8
