#!/bin/bash
# usage: tools/build_variants.sh <file.hip> NAME=-DFLAG[,-DFLAG2] ...   -> tools/variants/libspiral_NAME.so (ablation builds of one TU, with
# -DSPIRAL_TUNING: the threshold switches SPIRAL_MAC_*_MIN, SPIRAL_S2M_WIDE_MIN, SPIRAL_SWEEP_STAGE, SPIRAL_GRAPH_DOT exist only in these builds)
set -e
cd "$(dirname "$0")/../spiral_amd/csrc"
src=$1; shift
mkdir -p ../../tools/variants
objs=$(ls *.o | grep -v "^${src%.*}.o$")
for v in "$@"; do
  name=${v%%=*}; flags=$(echo "${v#*=}" | tr ',' ' ')
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -DSPIRAL_TUNING $flags -x hip -c $src -o /tmp/variant_$name.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../../tools/variants/libspiral_$name.so /tmp/variant_$name.o $objs
  echo built $name
done
