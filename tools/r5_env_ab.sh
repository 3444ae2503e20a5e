# same-box A/B of the training step with / without an environment switch: tools/r5_env_ab.sh VAR=VALUE
for rep in 1 2 3; do
  for v in off on; do
    if [ $v = on ]; then export "$1"; else unset "${1%%=*}"; fi
    python bench.py --no-workloads --no-cpu-baseline --no-alt --sustained 150 --steps 10 2>/dev/null | grep "^{" | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1 $v rep $rep: %.3f ms/step in the line, sustained %.3f ms (%.1f img/s), host enqueue %.2f' % (d['ms_per_step'], d['sustained']['ms_per_step'], d['sustained']['value'], d.get('host_enqueue_ms_per_step', 0)))"
  done
done
