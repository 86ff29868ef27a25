#!/bin/bash
# run on the GPU box: bench every variant under build/var plus the product lib; prints the stage times
ARGS="$@"
for lib in atsc_amd/lib/libatsc_hip.so build/var/*/libatsc_hip.so; do
  ATSC_HIP_LIB=$PWD/$lib timeout -k 10 120 python bench.py $ARGS --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d['stage_ms']; print('$lib', 'step', d['ms_per_step'], 'fir', s['mix_fir'], 'il', s['interleave'], 'rs', s['rand_rs'], 'osc', s['osc_table'])" || exit 1
done
