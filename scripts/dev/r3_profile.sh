#!/bin/bash
# GPU box: round-3 evidence run -- microbenchmark outputs kept as text, then the full rocprof profile of bench.py
set -o pipefail
mkdir -p gpurun_out/ub
timeout -k 10 60 scripts/ubench/bin/mfma_round > gpurun_out/ub/ubench_mfma_round.txt 2>&1 || { echo "mfma_round failed"; exit 1; }
cat gpurun_out/ub/ubench_mfma_round.txt
timeout -k 10 120 scripts/ubench/bin/store_dword_rows > gpurun_out/ub/ubench_store_dword_rows.txt 2>&1 || { echo "store_dword_rows failed"; exit 1; }
timeout -k 10 120 scripts/ubench/bin/valu_mfma2 > gpurun_out/ub/ubench_valu_mfma2.txt 2>&1 || { echo "valu_mfma2 failed"; exit 1; }
timeout -k 10 200 bash scripts/dev/power_probe2.sh > gpurun_out/ub/power_probe_wide_vs_fused.txt 2>&1 || { echo "power probe failed"; exit 1; }
cat gpurun_out/ub/power_probe_wide_vs_fused.txt
bash scripts/profile_gpu.sh r03
