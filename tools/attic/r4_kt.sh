R=$PWD; cd /tmp; export TMPDIR=/tmp
for W in "32 L/14" "256 B/32"; do
tag=$(echo $W | tr ' /' '__')
rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_$tag -o k -- python $R/tools/perf_clip.py $W > /dev/null 2>&1
n=180; [ "$W" = "256 B/32" ] && n=90
python $R/tools/trace_gaps.py $(find /tmp/kt_$tag -name "*kernel_trace.csv" | head -1) $n -q
done
