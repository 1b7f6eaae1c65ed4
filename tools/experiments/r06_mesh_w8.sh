# mesh decoder kernels with 8 waves per workgroup (two per SIMD, -DMD_WAVES=8: pdfnet_amd/libpdfnet_hip_w8.so) against the shipped 4: parity tests,
# the level bench, the step
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
out=gpurun_out/r06_mesh_w8.txt
: > $out
for lib in libpdfnet_hip.so libpdfnet_hip_w8.so; do
  echo "== $lib" >> $out
  PDFNET_HIP_LIB=$GRAFT_REPO_ROOT/pdfnet_amd/$lib timeout 600 python tools/mesh_bench.py 32 2>&1 | grep -v "^W\|amdgpu.ids" >> $out
done
echo "== parity on libpdfnet_hip_w8.so" >> $out
PDFNET_HIP_LIB=$GRAFT_REPO_ROOT/pdfnet_amd/libpdfnet_hip_w8.so timeout 900 python -m pytest tests/test_meshdec_gpu.py tests/test_dualgraph_golden_gpu.py -x -q 2>&1 | tail -4 >> $out
for r in 1 2; do
for lib in libpdfnet_hip.so libpdfnet_hip_w8.so; do
  echo "round $r $lib: img/s, ms/step" >> $out
  PDFNET_HIP_LIB=$GRAFT_REPO_ROOT/pdfnet_amd/$lib timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-mpjpe --no-roofline --no-bf16-legs --no-native-leg 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" >> $out 2>&1
done
done
cat $out
