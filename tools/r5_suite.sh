# Round 5: the GPU suite + one default-sized bench line on the same box
export MIOPEN_LOG_LEVEL=1
mkdir -p gpurun_out/r5a
timeout 2700 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 1500 -x 2>&1 | tail -25 > gpurun_out/r5a/gputest.log
cat gpurun_out/r5a/gputest.log
timeout 600 python bench.py --no-workloads --no-cpu-baseline --no-alt > gpurun_out/r5a/bench.log 2>&1
grep "^{" gpurun_out/r5a/bench.log | tail -1 | head -c 1500
