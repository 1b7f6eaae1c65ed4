# the fused mesh decoder's linear products as x3 arithmetic (csrc/meshdec_x3.hip, default) against the native fp32 MFMA build (PDF_X3_MESH=0): parity in both
# modes, the level bench, the step
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
out=gpurun_out/r06_mesh_x3.txt
: > $out
timeout 900 python -m pytest tests/test_meshdec_gpu.py tests/test_dualgraph_golden_gpu.py -x -q -s 2>&1 | grep "max |fused\|dx max\|passed\|failed\|Error\|error" | cut -c1-200 >> $out
for m in 0 1; do
  echo "== PDF_X3_MESH=$m" >> $out
  PDF_X3_MESH=$m timeout 600 python tools/mesh_bench.py 32 2>&1 | grep -v "^W\|amdgpu.ids" | grep "fused" >> $out
done
for r in 1 2; do for m in 0 1; do
  echo "round $r PDF_X3_MESH=$m: img/s, ms/step" >> $out
  PDF_X3_MESH=$m timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-mpjpe --no-roofline --no-bf16-legs --no-native-leg 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" >> $out 2>&1
done; done
timeout 900 python -m pytest tests/test_headline_gpu.py tests/test_model_gpu.py -x -q 2>&1 | tail -2 >> $out
cat $out
