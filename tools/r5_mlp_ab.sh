# same-box A/B of MLP-chain library variants (tools/build_mlp_variant.sh): training render (fwd+bwd), eval render, SDF backward alone; then the render parity tests on each variant
R=$PWD; mkdir -p gpurun_out/r5c
for rep in 1 2; do
  for v in "$@"; do
    if [ "$v" = base ]; then unset SHAPECLIPPER_HIP_LIB; else export SHAPECLIPPER_HIP_LIB=$R/shapeclipper_amd/lib/variants/lib_mlp_$v.so; fi
    timeout 200 python tools/perf_render.py --B 32 --iters 10 2>&1 | grep "ms per" | sed "s/^/$v rep $rep: /"
    timeout 200 python tools/perf_render.py --B 8 --full 128 --iters 3 2>&1 | grep "ms per" | sed "s/^/$v rep $rep: /"
    timeout 120 python tools/perf_sdf_bwd.py fused 2>&1 | grep "ms" | sed "s/^/$v rep $rep: /"
  done
done | tee gpurun_out/r5c/mlp_ab_$1_$2.txt
for v in "$@"; do
  [ "$v" = base ] && continue
  SHAPECLIPPER_HIP_LIB=$R/shapeclipper_amd/lib/variants/lib_mlp_$v.so timeout 1500 python -m pytest tests/test_gpu_sdf.py tests/test_gpu_sdf_backward.py tests/test_gpu_render_train.py tests/test_gpu_render_eval.py tests/test_gpu_parity_large.py tests/test_gpu_determinism.py tests/test_gpu_full_step_parity.py tests/test_gpu_render_hits.py tests/test_gpu_arch_variants.py tests/test_gpu_weight_norm.py tests/test_gpu_render_cabi.py -q -p no:cacheprovider 2>&1 | tail -6 | sed "s/^/$v: /"
done | tee -a gpurun_out/r5c/mlp_ab_$1_$2.txt
