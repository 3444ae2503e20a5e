# kernel trace + PMC view of one conv3x3 shape: bash tools/prof_conv.sh SIDE CH BATCH TAG
R=$PWD; O=$R/gpurun_out; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_ck_$4 -o k -- python $R/tools/perf_conv3x3.py $1 $2 $3 20 > $O/conv_prof_$4.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU --output-format csv -d /tmp/p_cv_$4 -o s -- python $R/tools/perf_conv3x3.py $1 $2 $3 3 >> $O/conv_prof_$4.log 2>&1
cd $R
python tools/summarize_prof.py $(find /tmp/p_ck_$4 -name "*kernel_stats.csv" | head -1) 6 >> $O/conv_prof_$4.log
python tools/summarize_pmc.py $(find /tmp/p_cv_$4 -name "*counter_collection.csv" | head -1) >> $O/conv_prof_$4.log
