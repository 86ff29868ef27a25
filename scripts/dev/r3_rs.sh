#!/bin/bash
set -o pipefail
O=gpurun_out/rs; mkdir -p $O
timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 200 --timeout-method thread -k "stage or rs or full_size or golden" > $O/t.log 2>&1 || { tail -20 $O/t.log; exit 1; }
tail -2 $O/t.log
for ppl in 1 2; do for grid in 1024 512 256; do
  ATSC_DEV_RS_PPL=$ppl ATSC_DEV_RS_GRID=$grid timeout -k 10 120 python bench.py --no-cpu-baseline --no-extras --steps 30 > $O/b_${ppl}_$grid.json 2> $O/b.err || { tail -5 $O/b.err; exit 1; }
  python -c "import json;d=json.load(open('$O/b_${ppl}_$grid.json'));print('ppl',$ppl,'grid',$grid,d['ms_per_step'],d['stage_ms']['rand_rs'],d['stage_ms']['interleave'],d['stage_ms']['mix_fir'])"
done; done
ATSC_DEV_RS_PPL=2 ATSC_DEV_RS_GRID=256 timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 200 --timeout-method thread -k "stage or rs or full_size or golden" > $O/t2.log 2>&1 || { tail -20 $O/t2.log; exit 1; }
tail -2 $O/t2.log
