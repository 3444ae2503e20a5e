# kernel-level profile of the meshing workload (marching cubes of 32 grids at 101^3): bash tools/prof_mc.sh
R=$PWD; mkdir -p $R/gpurun_out/prof_mc
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_mc -o mc -- python $R/tools/perf_mc.py > $R/gpurun_out/prof_mc/run.log 2>&1
cd $R
f=$(find /tmp/p_mc -name "*kernel_stats.csv" | head -1)
python tools/summarize_prof.py $f 14 | tee gpurun_out/prof_mc/summary.txt
