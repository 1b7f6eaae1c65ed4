cd ${GRAFT_REPO_ROOT:-$PWD}
timeout 300 python tools/x3_bench.py deconv > gpurun_out/r06_x3_deconv4.txt 2>&1
timeout 900 python -m pytest tests/test_headline_gpu.py -x -q -m gpu -s -k "pyramid_transposed" 2>&1 | grep -v "^W\|amdgpu.ids" | tail -25 >> gpurun_out/r06_x3_deconv4.txt
b="python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-mpjpe --no-bf16-legs --no-roofline --no-collective-path"
for cfg in "0" "1"; do
  v=$(PDFNET_X3_DECONV=$cfg timeout 300 $b 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['median_step_ms'])")
  echo "PDFNET_X3_DECONV=$cfg  img/s, ms/step, median ms: $v" >> gpurun_out/r06_x3_deconv4.txt
done
