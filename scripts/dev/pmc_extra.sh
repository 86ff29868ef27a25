#!/bin/bash
# Run ON THE GPU BOX: one extra SQ counter pass (stall attribution).  Usage: scripts/pmc_extra.sh <tag> <counters...>
TAG=$1; shift
OUT=$PWD/gpurun_out/pmcx_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python3 $PWD/bench.py --steps 5 --warmup 2 --no-cpu-baseline"
cd /tmp
timeout -s KILL 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT -- $BENCH > $OUT/log.txt 2>&1 || { echo failed; tail -5 $OUT/log.txt; exit 1; }
