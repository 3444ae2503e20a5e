export MIOPEN_LOG_LEVEL=1
python -m pytest "tests/test_gpu_fused_block.py" -x -q -p no:cacheprovider 2>&1 | grep -E "Error|assert|max|passed|failed" | head -12
for B in 32 16; do for F in "" "--opt=--hip.fused_block!"; do
python bench.py --batch $B --steps 30 --warmup 5 --no-cpu-baseline --no-workloads --sustained 0 --no-alt $F 2>/dev/null | grep "^{" | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('B=$B $F', {k:d[k] for k in ('value','ms_per_step','host_enqueue_ms_per_step')})"
done; done
