# Same-box A/B of the training step: baseline library build (argument 1, a path under shapeclipper_amd/lib/variants) against the tree's library.
# Alternates the two (A B A B) so that box drift shows; prints the sustained ms per step of each run.
R=$PWD; mkdir -p gpurun_out/r5g
for rep in 1 2; do
  for v in base new; do
    if [ $v = base ]; then export SHAPECLIPPER_HIP_LIB=$R/$1; else unset SHAPECLIPPER_HIP_LIB; fi
    python bench.py --no-workloads --no-cpu-baseline --no-alt --sustained 150 --steps 10 2>/dev/null | grep "^{" | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v rep $rep: %.3f ms/step in the line, sustained %.3f ms (%.1f img/s)' % (d['ms_per_step'], d['sustained']['ms_per_step'], d['sustained']['value']))"
  done
done | tee gpurun_out/r5g/ab_$(date +%H%M).txt
