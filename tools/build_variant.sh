#!/bin/bash
# A/B builds of the library: tools/build_variant.sh NAME "-DFLAG ..." -> ab/libiiv_NAME.so (iiv_greedy.hip rebuilt with the flags)
set -e
name="$1"; flags="$2"; src="${3:-iiv_greedy}"
cd "$(dirname "$0")/../ii-vision_amd/csrc"
mkdir -p ../../ab
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -mllvm -amdgpu-atomic-optimizer-strategy=None $flags -c $src.hip -o /tmp/${src}_$name.o
objs=""
for o in iiv_api iiv_tables iiv_bitmap iiv_encode iiv_prologue iiv_workgroup iiv_greedy iiv_team iiv_a2m iiv_ingest; do
  if [ "$o" = "$src" ]; then objs="$objs /tmp/${src}_$name.o"; else objs="$objs $o.o"; fi
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../../ab/libiiv_$name.so $objs
echo built ab/libiiv_$name.so
