#!/bin/bash
# profiling builds of the CLIP cluster kernel: tools/build_clip_prof.sh <name> [-DMACRO=v ...] -> shapeclipper_amd/lib/variants/lib_clip_<name>.so (SC_CL_PROF=1 always)
set -e
cd "$(dirname "$0")/../shapeclipper_amd/csrc"
name=$1; shift
mkdir -p ../lib/variants build/var
others=$(ls build/*.o | grep -v "build/clip_vit.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -I../../include -DSC_CL_PROF=1 "$@" -c clip_vit.hip -o build/var/clip_vit_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/variants/lib_clip_$name.so $others build/var/clip_vit_$name.o
