export MIOPEN_LOG_LEVEL=1
R=$PWD
O=$R/gpurun_out/r03_final
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 900 2>&1 | grep -E "passed|failed|Error" | tail -3 > $O/gputest.log
timeout 200 python tools/perf_bn.py 2>&1 | grep -v -i "warn\|amdgpu" > $O/bn_layers.txt
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_wl -o wl -- python $R/bench.py --workloads-only --no-cpu-baseline > $O/workloads_under_rocprof.log 2>&1
cd $R
cp $(find /tmp/p_wl -name "*kernel_stats.csv" | head -1) $O/workloads_kernel_stats.csv
python tools/summarize_prof.py $O/workloads_kernel_stats.csv 30 > $O/workloads_summary.txt
grep "^{" $O/workloads_under_rocprof.log | tail -1 > $O/workloads_line_under_rocprof.json; rm -f $O/workloads_under_rocprof.log
timeout 900 python bench.py > $O/bench_full.log 2>&1
grep "^{" $O/bench_full.log | tail -1 > $O/bench_line.json
tail -3 $O/bench_full.log | cut -c1-300
cat $O/gputest.log
