# x3gemm_tn tile variants on the weight-gradient shapes, then the step with the new automatic tile choice
cd ${GRAFT_REPO_ROOT:-$PWD}
timeout 300 python tools/x3_bench.py wgrad --nprod=6 --variants=0,2,3 2>&1 | grep "native\|x3"
timeout 300 python tools/x3_bench.py deconv 2>&1 | grep -v "^W\|amdgpu"
b="python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-mpjpe --no-bf16-legs --no-roofline --no-native-leg --no-collective-path"
for r in 1 2; do
  v=$(timeout 300 $b 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['median_step_ms'])")
  echo "round $r  step, img/s, ms/step, median ms: $v"
done
