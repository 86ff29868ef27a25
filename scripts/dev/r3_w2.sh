#!/bin/bash
set -o pipefail
O=$PWD/gpurun_out/r3d
mkdir -p $O
timeout -k 10 600 python -u -m pytest tests/test_gpu_round3.py -m gpu -x -q --timeout 180 --timeout-method thread -k "wide" -s > $O/tests.log 2>&1; rc=$?; tail -8 $O/tests.log; [ $rc -eq 0 ] || exit $rc
for abl in 0 1 2 3; do
  ATSC_DEV_ABL=$abl timeout -k 10 120 python bench.py --no-extras --no-cpu-baseline --steps 20 > $O/abl$abl.json 2> $O/abl$abl.err || { echo "abl $abl failed"; tail -3 $O/abl$abl.err; exit 1; }
  python -c "import json;d=json.load(open('$O/abl$abl.json'));print('abl',$abl,d['ms_per_step'],d['stage_ms']['mix_fir'])"
done
timeout -k 10 300 python bench.py --no-extras --no-cpu-baseline --steps 20 --fir-form fused > $O/bench_fused.json 2> $O/bench_fused.err; python -c "import json;d=json.load(open('$O/bench_fused.json'));print('fused',d['ms_per_step'],d['stage_ms']['mix_fir'])"
