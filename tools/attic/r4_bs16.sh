# Is the bs16 step (BASELINE config[1]) GPU-bound?  Sum of kernel durations per step from a rocprofv3 kernel trace, beside the step time
# measured without the profiler.
R=$PWD; O=$R/gpurun_out/r4_bs16; mkdir -p $O
python bench.py --batch 16 --no-workloads --no-cpu-baseline --no-alt --sustained 100 --steps 10 2>/dev/null | grep "^{" | tail -1 > $O/bench_bs16.json
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/p16
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p16 -o b -- python $R/bench.py --batch 16 --no-workloads --no-cpu-baseline --no-alt --sustained 0 --steps 10 > /dev/null 2>&1
cd $R
python tools/summarize_prof.py $(find /tmp/p16 -name "*kernel_stats.csv" | head -1) 25 > $O/bs16_kernel_summary.txt
python tools/gpu_busy.py $(find /tmp/p16 -name "*kernel_trace.csv" | head -1) 8 | head -8 > $O/bs16_gpu_busy.txt
python - <<PY
import json
d=json.load(open("$O/bench_bs16.json")); print("bs16 without profiler: %.3f ms in the line, sustained %.3f ms" % (d["ms_per_step"], d["sustained"]["ms_per_step"]))
PY
head -3 $O/bs16_kernel_summary.txt; cat $O/bs16_gpu_busy.txt
