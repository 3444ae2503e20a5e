# same-box A/B of the training step with / without an option: tools/r5_opt_ab.sh --hip.some_option!
for rep in 1 2 3; do
  for v in default "$1"; do
    if [ "$v" = default ]; then o=""; else o="--opt=$v"; fi
    python bench.py --no-workloads --no-cpu-baseline --no-alt --sustained 150 --steps 10 $o 2>/dev/null | grep "^{" | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v rep $rep: %.3f ms/step in the line, sustained %.3f ms (%.1f img/s), host enqueue %.2f' % (d['ms_per_step'], d['sustained']['ms_per_step'], d['sustained']['value'], d.get('host_enqueue_ms_per_step', 0)))"
  done
done
