mkdir -p gpurun_out/cl
for v in ab17 ab33 ab49; do
  echo "== $v"
  SHAPECLIPPER_HIP_LIB=$PWD/shapeclipper_amd/lib/variants/lib_clip_$v.so timeout 300 python tools/prof_clip_cluster.py 32 > gpurun_out/cl/prof_${v}_32.log 2>&1
  tail -1 gpurun_out/cl/prof_${v}_32.log
done
