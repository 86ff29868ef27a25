#!/bin/bash
# GPU box: the round's closing evidence -- full GPU suite, the driver's bench command, full profile
set -o pipefail
O=gpurun_out/final; mkdir -p $O
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $O/pytest_gpu.log 2>&1 || { tail -30 $O/pytest_gpu.log; exit 1; }
tail -2 $O/pytest_gpu.log
timeout -k 10 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err || { tail -5 $O/bench_driver_cmd.err; exit 1; }
wc -c $O/bench_driver_cmd.json
timeout -k 10 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err || { tail -5 $O/bench_default.err; exit 1; }
python - <<'PY'
import json
for n in ('bench_driver_cmd','bench_default'):
    d=json.load(open(f'gpurun_out/final/{n}.json'))
    print(n, d['value'], d['ms_per_step'], d['roofline']['frac'], d['stage_ms'], d['roofline']['iq_max_abs_error_vs_oracle']['max_abs_error'], d['sustained']['ms_per_step'], d['fp32_form']['ms_per_step'])
PY
bash scripts/profile_gpu.sh r03
