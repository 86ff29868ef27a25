#!/bin/bash
# GPU box: the round's closing evidence -- full GPU suite, default bench line, multi-stream null-driver throughput, full profile
set -o pipefail
O=gpurun_out/final; mkdir -p $O
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $O/pytest_gpu.log 2>&1 || { tail -30 $O/pytest_gpu.log; exit 1; }
tail -2 $O/pytest_gpu.log
timeout -k 10 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err || { tail -5 $O/bench_default.err; exit 1; }
python - <<'PY'
import json
d=json.load(open('gpurun_out/final/bench_default.json'))
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['stage_ms'], d.get('iq_max_abs_error_vs_oracle'), d['sustained']['ms_per_step'], d['fp32_form']['ms_per_step'], d.get('cpu_baseline'))
PY
# multi-stream front through the C++ group adapter: 128 fields of TS per stream, K copies
python - <<'PY'
import sys, numpy as np
sys.path.insert(0, 'tests')
from oracle_lib import ts_mt
ts_mt(4, 128).tofile('/tmp/in128.ts')
PY
for cfg in "1 8 cf32" "32 4 cf32" "64 4 cf32" "64 4 cs16" "64 8 cs8"; do
  set -- $cfg
  timeout -k 10 200 atsc_amd/bin/atsc_encoder_null /tmp/in128.ts $2 0 $3 --streams $1 > $O/null_$1_$2_$3.json 2> $O/null.err || { echo "null driver $cfg failed"; tail -3 $O/null.err; exit 1; }
  cat $O/null_$1_$2_$3.json
done
bash scripts/profile_gpu.sh r03
