# tile of the transposed convolutions' weight-gradient kernel (PDF_X3_TN_DECONV), isolated and in the step
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
out=gpurun_out/r06_tn_deconv.txt
: > $out
for v in 0 3 2; do
  echo "== PDF_X3_TN_DECONV=$v" >> $out
  PDF_X3_TN_DECONV=$v timeout 600 python tools/x3_bench.py deconv 2>&1 | grep -v "^W\|amdgpu.ids" | grep "x3 True" >> $out
done
PDF_X3_TN_DECONV=3 timeout 600 python -m pytest tests/test_headline_gpu.py -q -k "transposed" 2>&1 | tail -2 >> $out
for r in 1 2; do for v in 0 3 2; do
  echo "round $r PDF_X3_TN_DECONV=$v: img/s, ms/step" >> $out
  PDF_X3_TN_DECONV=$v timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-mpjpe --no-roofline --no-bf16-legs --no-native-leg 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" >> $out 2>&1
done; done
cat $out
