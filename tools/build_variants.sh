#!/bin/bash
# A/B builds of one kernel file: tools/build_variants.sh <file.hip> <MACRO> <v1> [v2 ...] -> shapeclipper_amd/lib/variants/lib_<MACRO>_<v>.so
# (select with SHAPECLIPPER_HIP_LIB=<path>; the directory is git-ignored and travels to the GPU box)
set -e
cd "$(dirname "$0")/../shapeclipper_amd/csrc"
make -s >/dev/null
f=$1; m=$2; shift 2
mkdir -p ../lib/variants build/var
base=$(basename $f .hip)
others=$(ls build/*.o | grep -v "build/$base.o")
for v in "$@"; do
  extra=""; [ "$base" = render ] && extra="-ffp-contract=off"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -I../../include $extra -D$m=$v -c $f -o build/var/${base}_$v.o &
done
wait
for v in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/variants/lib_${m}_$v.so $others build/var/${base}_$v.o
done
ls -la ../lib/variants/
