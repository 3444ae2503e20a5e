# SQ counters of the CLIP tower's kernels (north_star: MFMA utilisation for the ViT): one --pmc pass per workload, no tracing.
# Usage (GPU box): bash tools/prof_clip_pmc.sh ; summaries under gpurun_out/clip_pmc/
R=$PWD; O=$R/gpurun_out/clip_pmc; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
C="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU"
i=0
for W in "32 B/32" "256 B/32" "32 L/14"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $C --output-format csv -d /tmp/p_clip$i -o c -- python $R/tools/perf_clip.py $W > $O/run$i.log 2>&1
  echo "== ViT-$(echo $W | cut -d' ' -f2), batch $(echo $W | cut -d' ' -f1): largest dispatch of each kernel; MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (32 x SQ_BUSY_CYCLES) ==" >> $O/pmc_clip_sq.txt
  python $R/tools/summarize_pmc.py $(find /tmp/p_clip$i -name "*counter_collection.csv" | head -1) | awk '{ m=0; b=0; for (i=1;i<=NF;i++) { if ($i=="mfma_busy_cyc") m=$(i+1); if ($i=="busy_cyc") b=$(i+1) } if (b>0) printf "%s  mfma_busy %.0f%%\n", $0, 100*m/(32*b); else print $0 }' >> $O/pmc_clip_sq.txt
  grep "ViT" $O/run$i.log >> $O/pmc_clip_sq.txt
done
cat $O/pmc_clip_sq.txt | cut -c1-260
