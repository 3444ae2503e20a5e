export MIOPEN_LOG_LEVEL=1
mkdir -p gpurun_out/r4e
python -m pytest tests/test_gpu_fused_block.py tests/test_gpu_conv.py tests/test_gpu_determinism.py tests/test_gpu_full_step_parity.py tests/test_gpu_train_step.py tests/test_gpu_two_ranks.py -x -q -p no:cacheprovider 2>&1 | tail -6 > gpurun_out/r4e/tests.log
cat gpurun_out/r4e/tests.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-workloads --sustained 100 2>/dev/null | grep "^{" | tail -1 > gpurun_out/r4e/bench.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/r4e/bench.json'))
print({k:d[k] for k in ('value','ms_per_step','host_enqueue_ms_per_step')}, d['sustained'], d['config1_bs16'], d['roofline']['frac'], d['fp32_mfma_convolutions'])
PY
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-workloads --sustained 0 --no-alt --opt=--hip.fused_block! 2>/dev/null | grep "^{" | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('fused_block OFF:', {k:d[k] for k in ('value','ms_per_step','host_enqueue_ms_per_step')}, d['config1_bs16'])"
