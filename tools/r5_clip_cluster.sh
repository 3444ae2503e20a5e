mkdir -p gpurun_out/cl
timeout 900 python tools/dbg_clip_cluster.py stress 2>&1 | grep -v amdgpu.ids
