R=$PWD; O=$R/gpurun_out/r02_profiles; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_bench -o bench -- python $R/bench.py --no-workloads --no-cpu-baseline --no-alt --sustained 0 --steps 10 > $O/bench_under_rocprof.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_split -o bench -- python $R/bench.py --no-workloads --no-cpu-baseline --no-alt --sustained 0 --steps 10 --opt=--hip.conv3x3_split > $O/bench_split_under_rocprof.log 2>&1
cd $R
cp $(find /tmp/p_bench -name "*kernel_stats.csv" | head -1) $O/bench_bs32_kernel_stats.csv
cp $(find /tmp/p_split -name "*kernel_stats.csv" | head -1) $O/bench_bs32_split_kernel_stats.csv
python tools/summarize_prof.py $O/bench_bs32_kernel_stats.csv 45 > $O/bench_bs32_summary.txt
python tools/summarize_prof.py $O/bench_bs32_split_kernel_stats.csv 45 > $O/bench_bs32_split_summary.txt
tail -1 $O/bench_under_rocprof.log > $O/bench_line_under_rocprof.json; rm -f $O/bench_under_rocprof.log $O/bench_split_under_rocprof.log
