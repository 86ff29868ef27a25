#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/pytest_gpu.log 2>&1 || { tail -30 gpurun_out/pytest_gpu.log; exit 1; }
tail -3 gpurun_out/pytest_gpu.log
timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras > gpurun_out/bench_c.json 2> gpurun_out/bench_c.err || { tail -5 gpurun_out/bench_c.err; exit 1; }
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_c.json'))
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['stage_ms'])
PY
