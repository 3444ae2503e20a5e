export SC_CONV_VARIANT=1
for e in "" "SC_CONV_G1=1" "" "SC_CONV_G1=1"; do for f in "--opt=--hip.conv3x3"; do env $e python bench.py --no-workloads --no-cpu-baseline --sustained 100 $f 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d[\"ms_per_step\"], d.get(\"sustained\"), \"variant 1 $e $f\")" >> gpurun_out/r02_conv.log; done; done
