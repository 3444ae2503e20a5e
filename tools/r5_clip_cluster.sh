mkdir -p gpurun_out/cl
export SC_DBG_OUT=gpurun_out/cl
timeout 1500 python -m pytest tests/test_gpu_clip.py -x -q -s 2>&1 | grep -v amdgpu.ids > gpurun_out/cl/test_clip.log
grep -i "cluster form\|passed\|failed\|error" gpurun_out/cl/test_clip.log | tail -30
timeout 900 python tools/dbg_clip_cluster.py 12 16 20 24 28 32 > gpurun_out/cl/dbg_sweep.log 2>&1
grep -v amdgpu.ids gpurun_out/cl/dbg_sweep.log | grep "ms"
