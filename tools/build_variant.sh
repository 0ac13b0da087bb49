#!/bin/bash
# A/B builds of the library: one or more sources recompiled with extra flags, linked with the tree's other objects into
# vector_line_quantization_amd/csrc/variants/libvlq_<name>.so (git-ignored; select with VLQ_LIB_PATH).
#   tools/build_variant.sh phases "-DVLQ_SCAN16_PHASES" scan16
set -e
name=$1; flags=$2; shift 2
cd "$(dirname "$0")/../vector_line_quantization_amd/csrc"
make -s -j4
mkdir -p variants
objs=""
for f in *.hip; do
  b=${f%.hip}
  [ -f $b.o ] || continue      # (only what the Makefile's source list built)
  hit=0
  for s in "$@"; do [ "$s" = "$b" ] && hit=1; done
  if [ $hit = 1 ]; then
    extra=""
    [ "$b" = line16r ] && extra="-fno-slp-vectorize"
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function $extra $flags -c $f -o variants/${b}_$name.o
    objs="$objs variants/${b}_$name.o"
  else
    objs="$objs $b.o"
  fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/libvlq_$name.so $objs
echo "built variants/libvlq_$name.so"
