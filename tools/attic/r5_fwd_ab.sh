# same-box A/B: tree library against shapeclipper_amd/lib/variants/lib_mlp_$1.so -- training render, evaluation render, level grid; then the render parity tests on the tree library
R=$PWD; mkdir -p gpurun_out/r5h
for rep in 1 2 3; do
  for v in base $1; do
    if [ "$v" = base ]; then unset SHAPECLIPPER_HIP_LIB; else export SHAPECLIPPER_HIP_LIB=$R/shapeclipper_amd/lib/variants/lib_mlp_$v.so; fi
    timeout 200 python tools/perf_render.py --B 32 --iters 20 2>&1 | grep "ms per" | sed "s/^/$v rep $rep: /"
    timeout 200 python tools/perf_render.py --B 8 --full 128 --iters 5 2>&1 | grep "ms per" | sed "s/^/$v rep $rep: /"
  done
done | tee gpurun_out/r5h/fwd_ab_$1.txt
unset SHAPECLIPPER_HIP_LIB
timeout 1500 python -m pytest tests/test_gpu_sdf.py tests/test_gpu_sdf_backward.py tests/test_gpu_render_train.py tests/test_gpu_render_eval.py tests/test_gpu_parity_large.py tests/test_gpu_determinism.py tests/test_gpu_full_step_parity.py tests/test_gpu_render_hits.py tests/test_gpu_arch_variants.py tests/test_gpu_render_cabi.py -q -p no:cacheprovider 2>&1 | tail -4 | tee -a gpurun_out/r5h/fwd_ab_$1.txt
