# same-box A/B of the training step with / without option(s): tools/r6_step_ab.sh "<opts of arm A>" "<opts of arm B>" [tag]
R=$PWD; mkdir -p gpurun_out/r6ab
for rep in 1 2; do
  for arm in A B; do
    if [ $arm = A ]; then o="$1"; else o="$2"; fi
    python bench.py --no-workloads --no-cpu-baseline --no-alt --sustained 150 --steps 10 $o 2>/dev/null | grep "^{" | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); h=d['hip_ms_per_step']; print('arm $arm ($o) rep $rep: %.3f ms/step in the line, sustained %.3f ms (%.1f img/s); rgb fwd %.3f, rgb bwd %.3f, sdf fwd %.3f, sdf bwd %.3f ms/step' % (d['ms_per_step'], d['sustained']['ms_per_step'], d['sustained']['value'], h.get('sc_rgb_composite_forward',0), h.get('sc_rgb_composite_backward',0), h.get('sc_sdf_forward',0), h.get('sc_sdf_backward_fused',0)))"
  done
done | tee gpurun_out/r6ab/step_ab_${3:-x}.txt
