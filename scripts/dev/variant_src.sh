#!/bin/bash
# usage: variant_src.sh NAME SRC "-Dflags": like variant.sh but compiles SRC in place of fir_fused.hip
set -e
ROOT=$(cd $(dirname $0)/../.. && pwd)
N=$1; SRC=$2; D=$3
mkdir -p $ROOT/build/var/$N
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$ROOT/include -I$ROOT/atsc_amd/csrc -Wall -Wno-unused-result -Wno-unused-value $D -c $SRC -o $ROOT/build/var/$N/fir_fused.o
OBJS=$(ls $ROOT/build/obj/*.o | grep -v fir_fused.hip.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS $ROOT/build/var/$N/fir_fused.o -o $ROOT/build/var/$N/libatsc_hip.so
echo built $N
