# same-box A/B of the training step with / without option(s): tools/r5_step_ab.sh "<opts of arm A>" "<opts of arm B>"   (each: space-separated --opt=... or "")
R=$PWD; mkdir -p gpurun_out/r5e
for rep in 1 2; do
  for arm in A B; do
    if [ $arm = A ]; then o="$1"; else o="$2"; fi
    python bench.py --no-workloads --no-cpu-baseline --no-alt --sustained 150 --steps 10 $o 2>/dev/null | grep "^{" | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('arm $arm ($o) rep $rep: %.3f ms/step in the line, sustained %.3f ms (%.1f img/s)' % (d['ms_per_step'], d['sustained']['ms_per_step'], d['sustained']['value']))"
  done
done | tee gpurun_out/r5e/step_ab_$(date +%H%M%S).txt
