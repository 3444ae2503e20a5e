#!/bin/bash
# A/B builds of the MLP chain kernels: tools/build_mlp_variant.sh <name> <flags...>  -> shapeclipper_amd/lib/variants/lib_mlp_<name>.so
# (every file that includes mlp_tile.hpp is recompiled with the flags; select with SHAPECLIPPER_HIP_LIB=<path>)
set -e
cd "$(dirname "$0")/../shapeclipper_amd/csrc"
make -s >/dev/null
name=$1; shift
mkdir -p ../lib/variants build/var_$name
files="sdf_fwd rgb_fwd rgb_bwd sdf_bwdw sdf_bwd wgrad"
others=$(ls build/*.o)
for f in $files; do
  others=$(echo "$others" | grep -v "build/$f.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -I../../include "$@" -c $f.hip -o build/var_$name/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/variants/lib_mlp_$name.so $others build/var_$name/*.o
ls -la ../lib/variants/lib_mlp_$name.so
