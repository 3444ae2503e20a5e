export MIOPEN_LOG_LEVEL=1
mkdir -p gpurun_out/r4c
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 1500 -x 2>&1 | tail -15 > gpurun_out/r4c/gputest.log
cat gpurun_out/r4c/gputest.log
