#!/bin/bash
# usage: scripts/dev/variant.sh NAME "-DEXP_FOO -DEXP_BAR" [source.hip [path/to/other/version/of/source.hip]]
#   -> build/var/NAME/libatsc_hip.so (only that source differs from the product library)
set -e
ROOT=$(cd $(dirname $0)/../.. && pwd)
N=$1; D=$2; F=${3:-fir_fused.hip}; SRC=${4:-$ROOT/atsc_amd/csrc/$F}
mkdir -p $ROOT/build/var/$N
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$ROOT/include -I$ROOT/atsc_amd/csrc -Wall -Wno-unused-result -Wno-unused-value $D -c $SRC -o $ROOT/build/var/$N/var.o
OBJS=$(ls $ROOT/build/obj/*.o | grep -v "/$F.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS $ROOT/build/var/$N/var.o -o $ROOT/build/var/$N/libatsc_hip.so
echo built $N
