# Baseline at session start: GPU suite, bench line, kernel trace of the bench step -> busy / idle analysis
export MIOPEN_LOG_LEVEL=1
R=$PWD
O=$R/gpurun_out/r4s3
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 1200 2>&1 | grep -E "passed|failed|Error" | tail -3 > $O/gputest.log
timeout 600 python bench.py --no-workloads --no-cpu-baseline --no-alt --sustained 100 --steps 10 2>/dev/null | grep "^{" | tail -1 > $O/bench_line.json
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/p_bench -o bench -- python $R/bench.py --no-workloads --no-cpu-baseline --no-alt --sustained 0 --steps 10 > /dev/null 2>&1
cd $R
python tools/gpu_busy.py $(find /tmp/p_bench -name "*kernel_trace.csv" | head -1) 8 > $O/gpu_busy.txt 2>&1
cat $O/gputest.log; head -c 400 $O/bench_line.json; echo; cat $O/gpu_busy.txt
