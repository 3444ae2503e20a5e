# Same-box A/B of two TREES (argument 1: directory holding the other checkout, with its own built library): bench line + bs16 leg, A B A B
R=$PWD; mkdir -p gpurun_out/r5g
for rep in 1 2; do
  for v in prev new; do
    if [ $v = prev ]; then cd $R/$1; else cd $R; fi
    python bench.py --no-workloads --no-cpu-baseline --sustained 150 --steps 10 2>/dev/null | grep "^{" | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d.get('config1_bs16') or {}
print('$v rep $rep: bs32 %.3f ms in the line, sustained %.3f ms (%.1f img/s), host enqueue %.2f | bs16 %.3f ms' % (d['ms_per_step'], d['sustained']['ms_per_step'], d['sustained']['value'], d.get('host_enqueue_ms_per_step', 0), c.get('ms_per_step', 0)))"
  done
done | tee $R/gpurun_out/r5g/abtree_$(date +%H%M).txt
