export MIOPEN_LOG_LEVEL=1
python -m pytest tests/test_gpu_train_step.py tests/test_gpu_full_step_parity.py tests/test_gpu_fused_block.py tests/test_gpu_determinism.py -x -q -p no:cacheprovider 2>&1 | tail -3
for B in 32 16; do for F in "" "--opt=--hip.rocblas!"; do
python bench.py --batch $B --steps 40 --warmup 5 --no-cpu-baseline --no-workloads --sustained 0 --no-alt $F 2>/dev/null | grep "^{" | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('B=$B $F', {k:d[k] for k in ('value','ms_per_step','host_enqueue_ms_per_step')})"
done; done
