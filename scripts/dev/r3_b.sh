#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/ub
timeout -k 10 60 scripts/ubench/bin/mfma_round > gpurun_out/ub/ubench_mfma_round.txt 2>&1 || { echo "mfma_round failed"; exit 1; }
cat gpurun_out/ub/ubench_mfma_round.txt
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/pytest_gpu.log 2>&1 || { tail -30 gpurun_out/pytest_gpu.log; exit 1; }
tail -3 gpurun_out/pytest_gpu.log
timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/bench_b.json 2> gpurun_out/bench_b.err || { tail -5 gpurun_out/bench_b.err; exit 1; }
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_b.json'))
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['stage_ms'], d.get('iq_max_abs_error_vs_oracle'), d.get('sustained'))
PY
