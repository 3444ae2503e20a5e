# A/B of conv3x3.hip builds (tools/build_variants.sh conv3x3.hip <MACRO> v...): per-layer times of the split forward kernels
# usage: bash tools/r4_convvar.sh <MACRO> v1 v2 ...
mkdir -p gpurun_out/r4s3
M=$1; shift
for v in "$@"; do
  echo "== $M=$v"
  SHAPECLIPPER_HIP_LIB=$PWD/shapeclipper_amd/lib/variants/lib_${M}_$v.so python tools/perf_conv_split.py 30 2>&1 | tail -9
done | tee gpurun_out/r4s3/convvar_$(date +%H%M).txt
