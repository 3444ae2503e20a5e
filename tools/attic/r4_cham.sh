#!/bin/bash
# Chamfer grid search on surfaces a given distance apart: the far-tile rule (SC_CHAMFER_GRID_NEAR, 0 = always walk) against all pairs,
# then the same-bits tests, then per-kernel times (rocprofv3) of the regimes where the walk does not pay
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4cham; mkdir -p $O
for near in 0 2 3 4; do
  echo "== SC_CHAMFER_GRID_NEAR=$near" >> $O/near_sweep.txt
  SC_CHAMFER_GRID_NEAR=$near python $R/tools/perf_chamfer_surface.py 1 2>&1 | grep radius >> $O/near_sweep.txt
done
echo "== B=8, default" >> $O/near_sweep.txt
python $R/tools/perf_chamfer_surface.py 8 2>&1 | grep radius >> $O/near_sweep.txt
echo "== B=8, SC_CHAMFER_GRID_NEAR=0" >> $O/near_sweep.txt
SC_CHAMFER_GRID_NEAR=0 python $R/tools/perf_chamfer_surface.py 8 2>&1 | grep radius >> $O/near_sweep.txt
cd $R && timeout 900 python -m pytest tests/test_gpu_chamfer_grid.py tests/test_gpu_chamfer_ref.py tests/test_gpu_chamfer.py -x -q 2>&1 | tail -5 > $O/tests.log
cd /tmp
for d in 0.2 0.4; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pc_$d -o c -- python $R/tools/perf_chamfer_surface.py 1 $d > $O/run_$d.log 2>&1
  cp $(find /tmp/pc_$d -name '*kernel_stats.csv' | head -1) $O/kernel_stats_after_$d.csv
done
cat $O/near_sweep.txt; cat $O/tests.log
