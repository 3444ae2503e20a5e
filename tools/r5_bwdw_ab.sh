# same-box A/B of sdf_bwdw variants: kernel alone (tools/perf_sdf_bwd.py) + parity tests of the variant
R=$PWD; mkdir -p gpurun_out/r5b
for rep in 1 2; do
  for v in "$@"; do
    SHAPECLIPPER_HIP_LIB=$R/shapeclipper_amd/lib/variants/lib_SC_BWDW_VARIANT_$v.so timeout 120 python tools/perf_sdf_bwd.py fused 2>&1 | grep "ms" | sed "s/^/variant $v rep $rep: /"
  done
done | tee gpurun_out/r5b/bwdw_ab.txt
for v in "$@"; do
  [ "$v" = 0 ] && continue
  SHAPECLIPPER_HIP_LIB=$R/shapeclipper_amd/lib/variants/lib_SC_BWDW_VARIANT_$v.so timeout 900 python -m pytest tests/test_gpu_sdf_backward.py tests/test_gpu_parity_large.py tests/test_gpu_determinism.py -q -x -p no:cacheprovider 2>&1 | tail -3 | sed "s/^/variant $v: /"
done | tee -a gpurun_out/r5b/bwdw_ab.txt
