# x3 with the small products in their own accumulator set (X3_SPLIT_ACC=1, the default build) against the single-accumulator build
# (pdfnet_amd/libpdfnet_hip_oneacc.so, -DX3_SPLIT_ACC=0): error tests, kernel times, the step
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
out=gpurun_out/r06_split_acc_ab.txt
: > $out
timeout 600 python -m pytest tests/test_x3_gpu.py -q -s -rA 2>&1 | grep -v "^W\|amdgpu.ids" > gpurun_out/r06_x3_tests.txt
grep "NT \|TN \|rms error\|passed\|failed" gpurun_out/r06_x3_tests.txt >> $out
for lib in libpdfnet_hip_oneacc.so libpdfnet_hip.so; do
  echo "== $lib: tools/x3_bench.py" >> $out
  PDFNET_HIP_LIB=$GRAFT_REPO_ROOT/pdfnet_amd/$lib timeout 600 python tools/x3_bench.py --variants=0,1 2>&1 | grep -v "^W\|amdgpu.ids" >> $out
  PDFNET_HIP_LIB=$GRAFT_REPO_ROOT/pdfnet_amd/$lib timeout 600 python tools/x3_bench.py deconv 2>&1 | grep -v "^W\|amdgpu.ids" >> $out
done
for r in 1 2; do
for lib in libpdfnet_hip_oneacc.so libpdfnet_hip.so; do
  echo "round $r $lib: img/s, ms/step" >> $out
  PDFNET_HIP_LIB=$GRAFT_REPO_ROOT/pdfnet_amd/$lib timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-mpjpe --no-roofline --no-bf16-legs --no-native-leg 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" >> $out 2>&1
done
done
timeout 900 python -m pytest tests/test_headline_gpu.py -x -q 2>&1 | tail -3 >> $out
cat $out
