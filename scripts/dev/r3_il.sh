#!/bin/bash
set -o pipefail
O=gpurun_out/il; mkdir -p $O
ATSC_HIP_LIB=$PWD/atsc_amd/lib/libatsc_hip_$1.so timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 200 --timeout-method thread -k "stage or full_size or golden or edge or fused" > $O/t.log 2>&1 || { tail -20 $O/t.log; exit 1; }
tail -2 $O/t.log
VARIANTS="$1" bash scripts/dev/r3_ab.sh
