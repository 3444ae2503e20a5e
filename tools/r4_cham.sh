#!/bin/bash
# Chamfer grid search on surfaces a given distance apart: per-kernel times (rocprofv3) of the regimes where the walk does not pay
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4cham; mkdir -p $O
python $R/tools/perf_chamfer_surface.py 1 > $O/surface_b1.txt 2>&1
for d in 0.1 0.2 0.4; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pc_$d -o c -- python $R/tools/perf_chamfer_surface.py 1 $d > $O/run_$d.log 2>&1
  cp $(find /tmp/pc_$d -name '*kernel_stats.csv' | head -1) $O/kernel_stats_$d.csv
done
tail -n 20 $O/surface_b1.txt
