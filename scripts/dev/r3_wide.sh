#!/bin/bash
# wide-form bring-up: its parity test, then a short bench of both fused forms
set -o pipefail
O=gpurun_out/r3b
mkdir -p $O
timeout -k 10 600 python -u -m pytest tests/test_gpu_round3.py tests/test_zz_no_alloc.py -m gpu -x -q --timeout 180 --timeout-method thread -k "wide or static" -s > $O/tests.log 2>&1; rc=$?; tail -15 $O/tests.log; [ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py --no-extras --no-cpu-baseline --steps 20 > $O/bench_wide.json 2> $O/bench_wide.err; rc=$?; echo "bench wide rc=$rc"; [ $rc -eq 0 ] || { tail -5 $O/bench_wide.err; exit $rc; }
timeout -k 10 300 python bench.py --no-extras --no-cpu-baseline --steps 20 --fir-form fused > $O/bench_fused.json 2> $O/bench_fused.err; echo "bench fused rc=$?"
python - <<'PY'
import json
for n in ("bench_wide","bench_fused"):
    d=json.load(open(f"gpurun_out/r3b/{n}.json")); print(n, d["value"], d["ms_per_step"], d["stage_ms"], d["roofline"]["frac"])
PY
