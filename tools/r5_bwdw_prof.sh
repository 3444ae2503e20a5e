R=$PWD; mkdir -p gpurun_out/r5b
for v in "$@"; do
  SC_BWDW_PROF_LIB=$R/tools/micro/libbwdw_prof_v$v.so timeout 300 python tools/prof_bwdw.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r5b/bwdw_phase_v$v.txt
  cat gpurun_out/r5b/bwdw_phase_v$v.txt
done
