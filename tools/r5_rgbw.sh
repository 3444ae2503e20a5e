mkdir -p gpurun_out/r5f
for rep in 1 2 3; do
  timeout 200 python tools/perf_render.py --B 32 --iters 20 '--opt=--hip.fused_rgb_wgrad!' 2>&1 | grep "ms per"
  timeout 200 python tools/perf_render.py --B 32 --iters 20 2>&1 | grep "ms per"
done | tee gpurun_out/r5f/rgb_fused_ab.txt
bash tools/r5_step_ab.sh "--opt=--hip.fused_rgb_wgrad!" ""
