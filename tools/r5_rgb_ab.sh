# same-box A/B of the training render: tree library against shapeclipper_amd/lib/variants/$1 (and option $2 on the tree library), then the phase profile and the parity tests
for rep in 1 2 3; do
  timeout 200 python tools/perf_render.py --B 32 --iters 20 2>&1 | grep "ms per" | sed "s/^/tree: /"
  SHAPECLIPPER_HIP_LIB=$PWD/shapeclipper_amd/lib/variants/$1 timeout 200 python tools/perf_render.py --B 32 --iters 20 2>&1 | grep "ms per" | sed "s/^/$1: /"
  [ -n "$2" ] && timeout 200 python tools/perf_render.py --B 32 --iters 20 --opt=$2 2>&1 | grep "ms per" | sed "s/^/tree: /"
done
SHAPECLIPPER_HIP_LIB=$PWD/shapeclipper_amd/lib/variants/lib_SC_RGBB_PROFILE_1.so timeout 300 python tools/prof_rgb_bwd.py 2>&1 | tail -21
timeout 900 python -m pytest tests/test_gpu_rgb_stash.py tests/test_gpu_render_train.py tests/test_gpu_parity_large.py tests/test_gpu_full_step_parity.py tests/test_gpu_determinism.py tests/test_gpu_render_cabi.py tests/test_gpu_render_hits.py -q -x 2>&1 | tail -4
