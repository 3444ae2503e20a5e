# Round-6 profile collection (run on the GPU box): rocprofv3 kernel stats of the bench step and of the other SURVEY 8(d) workloads, PMC passes
# (HBM bytes, SQ stalls) on one B=32 training render, SQ counters of the CLIP tower, the full bench line.  Only the small summaries are kept
# (gpurun_out/r06_profiles -> profiles/r06_*).  The GPU suite is run separately (tools/r6_suite.sh).
export MIOPEN_LOG_LEVEL=1
R=$PWD
O=$R/gpurun_out/r06_profiles
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_bench -o bench -- python $R/bench.py --no-workloads --no-cpu-baseline --no-alt --sustained 0 --steps 10 > $O/bench_under_rocprof.log 2>&1
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_wl -o wl -- python $R/bench.py --workloads-only --no-cpu-baseline > $O/workloads_under_rocprof.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/p_fetch -o f -- python $R/tools/perf_render.py --B 32 --iters 1 --yaml $R/options/pix3d/config.yaml > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/p_write -o w -- python $R/tools/perf_render.py --B 32 --iters 1 --yaml $R/options/pix3d/config.yaml > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU --output-format csv -d /tmp/p_sq -o s -- python $R/tools/perf_render.py --B 32 --iters 1 --yaml $R/options/pix3d/config.yaml > /dev/null 2>&1
cd $R
cp $(find /tmp/p_bench -name "*kernel_stats.csv" | head -1) $O/bench_bs32_kernel_stats.csv
cp $(find /tmp/p_wl -name "*kernel_stats.csv" | head -1) $O/workloads_kernel_stats.csv
python tools/summarize_prof.py $O/bench_bs32_kernel_stats.csv 60 > $O/bench_bs32_summary.txt
python tools/summarize_prof.py $O/workloads_kernel_stats.csv 30 > $O/workloads_summary.txt
python tools/summarize_pmc.py $(find /tmp/p_fetch -name "*counter_collection.csv" | head -1) > $O/pmc_fetch_size.txt
python tools/summarize_pmc.py $(find /tmp/p_write -name "*counter_collection.csv" | head -1) > $O/pmc_write_size.txt
python tools/summarize_pmc.py $(find /tmp/p_sq -name "*counter_collection.csv" | head -1) > $O/pmc_sq_stalls.txt
python tools/make_traffic_json.py $O/pmc_fetch_size.txt $O/pmc_write_size.txt > $O/traffic.json
grep "^{" $O/bench_under_rocprof.log | tail -1 > $O/bench_line_under_rocprof.json; rm -f $O/bench_under_rocprof.log
grep "^{" $O/workloads_under_rocprof.log | tail -1 > $O/workloads_line_under_rocprof.json; rm -f $O/workloads_under_rocprof.log
bash tools/prof_clip_pmc.sh > /dev/null 2>&1; cp gpurun_out/clip_pmc/pmc_clip_sq.txt $O/pmc_clip_sq.txt 2>/dev/null
timeout 900 python bench.py > $O/bench_full.log 2>&1
grep "^{" $O/bench_full.log | tail -1 > $O/bench_line.json; tail -5 $O/bench_full.log | grep -v "^{" | head -3; rm -f $O/bench_full.log
head -c 700 $O/bench_line.json; echo; head -30 $O/bench_bs32_summary.txt; cat $O/traffic.json | head -40; du -sh $O
