R=$PWD; O=$R/gpurun_out; cd /tmp && export TMPDIR=/tmp
for f in "--opt=--hip.conv3x3"; do
rm -rf /tmp/p_idle
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_idle -o t -- python $R/bench.py --no-workloads --no-cpu-baseline --sustained 0 --steps 10 $f > /dev/null 2>&1
echo "== $f" >> $O/r02_idle.log
python $R/tools/summarize_prof.py $(find /tmp/p_idle -name "*kernel_stats.csv" | head -1) 30 >> $O/r02_idle.log
done
