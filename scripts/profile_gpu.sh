#!/bin/bash
# Run ON THE GPU BOX (through gpurun): kernel-trace stats + PMC passes for bench.py.
# Usage: scripts/profile_gpu.sh <tag>      -> gpurun_out/prof_<tag>/...
set -o pipefail
TAG=${1:-r01}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python3 $PWD/bench.py --steps 5 --warmup 2 --no-cpu-baseline"
# the trace pass runs the default bench (50 steps after 10 warm-up steps) so that its per-kernel averages are those of bench.py's own line
BENCH_TRACE="python3 $PWD/bench.py --no-cpu-baseline"
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH_TRACE > $OUT/trace.log 2>&1 || { echo "trace failed"; tail -5 $OUT/trace.log; exit 1; }
# PMC passes: one counter block budget per pass (8 SQ / 4 TCC; FETCH_SIZE=3 TCC, WRITE_SIZE=2 TCC)
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $OUT/pmc_sq1 -- $BENCH > $OUT/pmc_sq1.log 2>&1 || { echo "pmc_sq1 failed"; tail -5 $OUT/pmc_sq1.log; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d $OUT/pmc_sq2 -- $BENCH > $OUT/pmc_sq2.log 2>&1 || { echo "pmc_sq2 failed"; tail -5 $OUT/pmc_sq2.log; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- $BENCH > $OUT/pmc_fetch.log 2>&1 || { echo "pmc_fetch failed"; tail -5 $OUT/pmc_fetch.log; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- $BENCH > $OUT/pmc_write.log 2>&1 || { echo "pmc_write failed"; tail -5 $OUT/pmc_write.log; exit 1; }
find $OUT -name "*.csv" | head -40
