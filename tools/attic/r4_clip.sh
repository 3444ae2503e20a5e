mkdir -p gpurun_out/r4b
python -m pytest tests/test_gpu_clip.py tests/test_cabi.py -x -q > gpurun_out/r4b/tests.log 2>&1; echo "tests rc $?" >> gpurun_out/r4b/tests.log
for rep in 1 2; do for W in "32 B/32" "256 B/32" "32 L/14"; do python tools/perf_clip.py $W 2>&1 | grep ViT; done; done > gpurun_out/r4b/perf.log 2>&1
tail -3 gpurun_out/r4b/tests.log; cat gpurun_out/r4b/perf.log
