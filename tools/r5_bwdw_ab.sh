# same-box A/B of sdf_bwdw variants of macro $1 (values $2...): kernel alone (tools/perf_sdf_bwd.py) + parity tests of each non-zero variant
R=$PWD; mkdir -p gpurun_out/r5b; M=$1; shift
for rep in 1 2; do
  for v in "$@"; do
    SHAPECLIPPER_HIP_LIB=$R/shapeclipper_amd/lib/variants/lib_${M}_$v.so timeout 120 python tools/perf_sdf_bwd.py fused 2>&1 | grep "ms" | sed "s/^/$M=$v rep $rep: /"
  done
done | tee gpurun_out/r5b/bwdw_ab_$M.txt
for v in "$@"; do
  [ "$v" = 0 ] && continue
  SHAPECLIPPER_HIP_LIB=$R/shapeclipper_amd/lib/variants/lib_${M}_$v.so timeout 900 python -m pytest tests/test_gpu_sdf_backward.py tests/test_gpu_parity_large.py tests/test_gpu_determinism.py tests/test_gpu_render_train.py -q -x -p no:cacheprovider 2>&1 | tail -3 | sed "s/^/$M=$v: /"
done | tee -a gpurun_out/r5b/bwdw_ab_$M.txt
