#!/bin/bash
set -o pipefail
O=gpurun_out/rs; mkdir -p $O
timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 200 --timeout-method thread -k "stage or rs or full_size or golden or edge" > $O/t.log 2>&1 || { tail -20 $O/t.log; exit 1; }
tail -2 $O/t.log
timeout -k 10 120 python bench.py --no-cpu-baseline --no-extras --steps 30 > $O/b.json 2> $O/b.err || { tail -5 $O/b.err; exit 1; }
python -c "import json;d=json.load(open('$O/b.json'));print(d['ms_per_step'],d['stage_ms'])"
