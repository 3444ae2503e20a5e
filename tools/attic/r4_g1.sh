mkdir -p gpurun_out/r4a
R=$PWD
python -m pytest tests/test_gpu_bench_contract.py::test_workloads_line tests/test_gpu_clip.py -x -q -m gpu > gpurun_out/r4a/tests.log 2>&1; echo "tests rc $?" >> gpurun_out/r4a/tests.log
cd /tmp && export TMPDIR=/tmp
for W in "32 B/32" "256 B/32" "32 L/14"; do
  tag=$(echo $W | tr ' /' '__')
  rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_$tag -o k -- python $R/tools/perf_clip.py $W > $R/gpurun_out/r4a/run_$tag.log 2>&1
  f=$(find /tmp/kt_$tag -name "*kernel_trace.csv" | head -1)
  n=90; [ "$W" = "32 L/14" ] && n=180
  python $R/tools/trace_gaps.py $f $n > $R/gpurun_out/r4a/gaps_$tag.txt 2>&1
done
cd $R
tail -3 gpurun_out/r4a/tests.log; tail -25 gpurun_out/r4a/gaps_32_B_32.txt
