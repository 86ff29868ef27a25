#!/bin/bash
# GPU box: same-box A/B of library variants (ATSC_HIP_LIB), default bench without the extras, two passes
set -o pipefail
O=gpurun_out/ab; mkdir -p $O
for pass in 1 2; do
for v in "" $VARIANTS; do
  lib=atsc_amd/lib/libatsc_hip${v:+_$v}.so
  ATSC_HIP_LIB=$PWD/$lib timeout -k 10 120 python bench.py --no-cpu-baseline --no-extras --steps 40 > $O/b_${v:-base}_$pass.json 2> $O/b.err || { echo "variant '$v' failed"; tail -5 $O/b.err; exit 1; }
  python -c "import json;d=json.load(open('$O/b_${v:-base}_$pass.json'));print('${v:-base}',$pass,d['ms_per_step'],d['stage_ms']['mix_fir'],d['stage_ms']['rand_rs'],d['stage_ms']['interleave'])"
done; done
