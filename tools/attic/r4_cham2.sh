#!/bin/bash
# all pairs with the slicing rule (old rule = SHAPECLIPPER_CHAMFER_NSPLIT=20 at b=1), the grid search's adaptive fallback slicing, same-bits tests
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4cham2; mkdir -p $O; rm -f $O/*.txt
echo "== B=1 auto" >> $O/sweep.txt;  python $R/tools/perf_chamfer_surface.py 1 2>&1 | grep radius >> $O/sweep.txt
echo "== B=1 nsplit=20 (round-3 rule)" >> $O/sweep.txt; SHAPECLIPPER_CHAMFER_NSPLIT=20 python $R/tools/perf_chamfer_surface.py 1 0.2 2>&1 | grep radius >> $O/sweep.txt
for s in 4 6; do echo "== B=1 auto, SC_CHAMFER_SLOTS_PER_CU=$s" >> $O/sweep.txt; SC_CHAMFER_SLOTS_PER_CU=$s python $R/tools/perf_chamfer_surface.py 1 0.2 2>&1 | grep radius >> $O/sweep.txt; done
echo "== B=8 auto" >> $O/sweep.txt;  python $R/tools/perf_chamfer_surface.py 8 2>&1 | grep radius >> $O/sweep.txt
echo "== B=32 auto (0.2 only)" >> $O/sweep.txt;  python $R/tools/perf_chamfer_surface.py 32 0.2 2>&1 | grep radius >> $O/sweep.txt
echo "== B=32 nsplit=1 (0.2 only)" >> $O/sweep.txt;  SHAPECLIPPER_CHAMFER_NSPLIT=1 python $R/tools/perf_chamfer_surface.py 32 0.2 2>&1 | grep radius >> $O/sweep.txt
cd $R && timeout 900 python -m pytest tests/test_gpu_chamfer_grid.py tests/test_gpu_chamfer_ref.py tests/test_gpu_chamfer.py -x -q 2>&1 | tail -5 > $O/tests.log
cat $O/sweep.txt; cat $O/tests.log
