# Round 6: the GPU suite (driver's command, -x) + the 8-rank config[3] test repeated + one default bench line, on one box
export MIOPEN_LOG_LEVEL=1
OUT=gpurun_out/${1:-r6a}
mkdir -p $OUT
timeout 3000 python -m pytest tests -m gpu -x -q -p no:cacheprovider --timeout 1500 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -30 > $OUT/gputest.log
cat $OUT/gputest.log
REPS=${2:-10}
: > $OUT/eight_ranks.log
for i in $(seq 1 $REPS); do
  timeout 900 python -m pytest tests/test_gpu_bench_contract.py -q -p no:cacheprovider -k "eight_ranks or two_ranks_through" 2>&1 | tail -1 >> $OUT/eight_ranks.log
done
cat $OUT/eight_ranks.log
timeout 900 python bench.py > $OUT/bench.log 2>&1
grep "^{" $OUT/bench.log | tail -1 > $OUT/bench.json
head -c 1200 $OUT/bench.json
