#!/bin/bash
# run on the GPU box: bench every variant under build/var plus the product lib; prints mix_fir ms
ARGS="$@"
for lib in atsc_amd/lib/libatsc_hip.so build/var/*/libatsc_hip.so; do
  ATSC_HIP_LIB=$PWD/$lib timeout -k 10 120 python bench.py $ARGS --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$lib', 'step', d['ms_per_step'], 'fir', d['stage_ms']['mix_fir'], 'map', d['stage_ms']['trellis_map'])" || exit 1
done
