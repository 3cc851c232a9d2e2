#!/bin/bash
# usage: tools/build_variants.sh <file.hip> NAME=-DFLAG[,-DFLAG2] ...   -> tools/variants/libspiral_NAME.so (ablation builds of one TU)
set -e
cd "$(dirname "$0")/../spiral_amd/csrc"
src=$1; shift
mkdir -p ../../tools/variants
objs=$(ls *.o | grep -v "^${src%.*}.o$")
for v in "$@"; do
  name=${v%%=*}; flags=$(echo "${v#*=}" | tr ',' ' ')
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function $flags -x hip -c $src -o /tmp/variant_$name.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../../tools/variants/libspiral_$name.so /tmp/variant_$name.o $objs
  echo built $name
done
