#!/bin/bash
# usage: scripts/dev/variant.sh NAME "-DEXP_FOO -DEXP_BAR"  -> build/var/NAME/libatsc_hip.so (only fir_fused.hip differs)
set -e
ROOT=$(cd $(dirname $0)/../.. && pwd)
N=$1; D=$2
mkdir -p $ROOT/build/var/$N
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$ROOT/include -I$ROOT/atsc_amd/csrc -Wall -Wno-unused-result -Wno-unused-value $D -c $ROOT/atsc_amd/csrc/fir_fused.hip -o $ROOT/build/var/$N/fir_fused.o
OBJS=$(ls $ROOT/build/obj/*.o | grep -v fir_fused.hip.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS $ROOT/build/var/$N/fir_fused.o -o $ROOT/build/var/$N/libatsc_hip.so
echo built $N
