#!/bin/bash
# round-3 first GPU call: MFMA subnormal probe, full GPU test suite, default bench, RCCL-backed one-rank time-split records
set -o pipefail
O=gpurun_out/r3a
mkdir -p $O
timeout -k 10 60 scripts/ubench/bin/mfma_denorm > $O/mfma_denorm.txt 2>&1; echo "denorm rc=$?"; cat $O/mfma_denorm.txt
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > $O/tests.log 2>&1; rc=$?; tail -5 $O/tests.log; [ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py > $O/bench_default.json 2> $O/bench_default.err; rc=$?; echo "bench rc=$rc"; [ $rc -eq 0 ] || { tail -5 $O/bench_default.err; exit $rc; }
timeout -k 10 300 python bench.py --mode timesplit --rccl --steps 30 > $O/bench_timesplit_1rank_rccl.json 2> $O/ts1.err; rc=$?; echo "ts rccl rc=$rc"; [ $rc -eq 0 ] || { tail -5 $O/ts1.err; exit $rc; }
timeout -k 10 300 python bench.py --mode timesplit --rccl --channels 64 --fields 16 --steps 30 > $O/bench_timesplit_64ch_1rank_rccl.json 2> $O/ts64.err; rc=$?; echo "ts64 rccl rc=$rc"; [ $rc -eq 0 ] || { tail -5 $O/ts64.err; exit $rc; }
timeout -k 10 300 python bench.py --mode timesplit --steps 30 > $O/bench_timesplit_1gpu_noproc.json 2> $O/ts0.err; echo "ts none rc=$?"
python - <<'PY'
import json
for n in ("bench_default","bench_timesplit_1rank_rccl","bench_timesplit_64ch_1rank_rccl","bench_timesplit_1gpu_noproc"):
    try:
        d=json.load(open(f"gpurun_out/r3a/{n}.json"))
        print(n, d["value"], d["ms_per_step"], d.get("backend"), d.get("parity"), d.get("handoff_latency_us_median"), d.get("sustained"), (d.get("fp32_form") or {}).get("ms_per_step"), d.get("roofline",{}).get("iq_max_abs_error_vs_oracle"), d.get("roofline",{}).get("frac"))
    except Exception as e: print(n, "unreadable", e)
PY
