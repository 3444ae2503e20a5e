#!/bin/bash
# guarded optimizer step (device-side skip + one-step-late host check) against the per-step host wait; tests first
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4guard; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_train_step.py tests/test_gpu_two_ranks.py tests/test_gpu_chamfer.py -x -q 2>&1 | tail -15 > $O/tests.log
F="--no-workloads --no-cpu-baseline --no-alt --sustained 0 --steps 100 --warmup 10"
for rep in 1 2; do
for b in 32 16; do
  python bench.py $F --batch $b 2>/dev/null | tail -1 > $O/b${b}_guarded_$rep.json
  python bench.py $F --batch $b --opt=--hip.guarded_step! 2>/dev/null | tail -1 > $O/b${b}_hostwait_$rep.json
done; done
python - <<'PY' | tee gpurun_out/r4guard/summary.txt
import json,glob
for f in sorted(glob.glob('gpurun_out/r4guard/*.json')):
    d=json.loads(open(f).read()); print(f.split('/')[-1], 'ms/step', d['ms_per_step'], 'host enqueue', d.get('host_enqueue_ms_per_step'))
PY
cat $O/tests.log
