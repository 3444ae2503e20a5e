#!/bin/bash
# product builds of the CLIP cluster kernel at other ring lengths: tools/build_clip_ring.sh 3 4 5 -> shapeclipper_amd/lib/variants/lib_clip_ring<N>.so
set -e
cd "$(dirname "$0")/../shapeclipper_amd/csrc"
mkdir -p ../lib/variants build/var
others=$(ls build/*.o | grep -v "build/clip_vit.o")
for n in "$@"; do
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -I../../include -DSC_CL_RING=$n -c clip_vit.hip -o build/var/clip_vit_ring$n.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/variants/lib_clip_ring$n.so $others build/var/clip_vit_ring$n.o ) &
done
wait
