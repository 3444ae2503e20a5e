#!/bin/bash
# What does the per-step host wait of the NaN / Inf check cost?  bs16 and bs32 with and without --check_finite
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4sync; mkdir -p $O
F="--no-workloads --no-cpu-baseline --no-alt --sustained 0 --steps 100 --warmup 10"
for b in 16 32; do
  python bench.py $F --batch $b 2>/dev/null | tail -1 > $O/b${b}_check.json
  python bench.py $F --batch $b --opt=--check_finite! 2>/dev/null | tail -1 > $O/b${b}_nocheck.json
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4sync/*.json')):
    d=json.loads(open(f).read()); print(f, d['ms_per_step'], d.get('host_enqueue_ms_per_step'))
PY
