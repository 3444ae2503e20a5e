# End-of-session verification on the GPU box: build check of the shipped library, full GPU suite, smoke(), the default bench line
export MIOPEN_LOG_LEVEL=1
mkdir -p gpurun_out/r4_final
timeout 1800 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 1200 2>&1 | grep -E "passed|failed|rror" | tail -3 > gpurun_out/r4_final/gputest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 > gpurun_out/r4_final/smoke.log
( time timeout 900 python bench.py > gpurun_out/r4_final/bench_full.log 2>&1 ) 2> gpurun_out/r4_final/bench_time.log
grep "^{" gpurun_out/r4_final/bench_full.log | tail -1 > gpurun_out/r4_final/bench_line.json; rm -f gpurun_out/r4_final/bench_full.log
cat gpurun_out/r4_final/gputest.log gpurun_out/r4_final/smoke.log gpurun_out/r4_final/bench_time.log; head -c 300 gpurun_out/r4_final/bench_line.json
