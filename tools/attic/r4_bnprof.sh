# GPU-side durations of the BatchNorm kernels per layer shape, one-launch form against two-launch form (lib variant SC_BN_NO_FUSED)
R=$PWD; mkdir -p $R/gpurun_out/r4s3
cd /tmp && export TMPDIR=/tmp
for v in fused nofused; do
  rm -rf /tmp/p_bn
  [ $v = nofused ] && export SHAPECLIPPER_HIP_LIB=$R/shapeclipper_amd/lib/variants/lib_SC_BN_NO_FUSED_1.so
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/p_bn -o bn -- python $R/tools/perf_bn.py > /dev/null 2>&1
  echo "== $v"
  python $R/tools/trace_by_grid.py $(find /tmp/p_bn -name "*kernel_trace.csv" | head -1) bn_
done | tee $R/gpurun_out/r4s3/bn_kernels.txt
