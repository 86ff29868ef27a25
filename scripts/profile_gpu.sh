#!/bin/bash
# Run ON THE GPU BOX (through gpurun): kernel-trace stats + PMC passes for bench.py.
# Usage: scripts/profile_gpu.sh <tag>      -> gpurun_out/prof_<tag>/...   (summarise with scripts/summarize_profile.py)
set -o pipefail
TAG=${1:-r03}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python3 $PWD/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras"
# the trace pass runs the default bench (its own step and warm-up counts) so that its per-kernel averages are those of bench.py's own line
BENCH_TRACE="python3 $PWD/bench.py --no-cpu-baseline"
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH_TRACE > $OUT/trace.log 2>&1 || { echo "trace failed"; tail -5 $OUT/trace.log; exit 1; }
grep -h '"metric"' $OUT/trace.log > $OUT/bench_under_trace.json
# PMC passes, one run each: at most 8 SQ / 4 TCC (FETCH_SIZE = 3, WRITE_SIZE = 2) / 4 TCP / 2 TA / 2 TD / 2 GRBM counters per pass
pass() { timeout -s KILL 120 rocprofv3 --pmc $2 --kernel-trace --output-format csv -d $OUT/pmc_$1 -- $BENCH > $OUT/pmc_$1.log 2>&1 || { echo "pmc_$1 failed"; tail -5 $OUT/pmc_$1.log; exit 1; }; }
pass sq1 "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY"
pass sq2 "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM"
pass sq3 "SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_ACTIVE_INST_LDS SQ_INSTS_VALU"
pass fetch "FETCH_SIZE GRBM_GUI_ACTIVE"
pass write "WRITE_SIZE"
pass l2 "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_WRREQ_STALL_sum"
pass tcp1 "TA_TA_BUSY_sum TD_TD_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"
pass tcp2 "TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum"
find $OUT -name "*.csv" | wc -l
