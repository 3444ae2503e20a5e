# GEMM lab on the GPU box: bash tools/r4_lab.sh [variants...]   (default: the plain binary)
mkdir -p gpurun_out/lab
T=$(date +%H%M%S)
for v in ${@:-lab}; do
  b=tools/micro/gemm_$v.bin
  echo "=== $v ===" >> gpurun_out/lab/lab_$T.txt
  timeout 300 $b >> gpurun_out/lab/lab_$T.txt 2>&1; echo "rc $?" >> gpurun_out/lab/lab_$T.txt
done
cat gpurun_out/lab/lab_$T.txt | cut -c1-150
