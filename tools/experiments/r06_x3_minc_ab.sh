# which F(4x4) layers should run their transform-domain products as x3 (PDF_X3_MINC channels, PDF_X3_MINT tiles)
cd ${GRAFT_REPO_ROOT:-$PWD}
b="python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-mpjpe --no-bf16-legs --no-roofline --no-native-leg --no-collective-path"
for r in 1 2; do
  for cfg in ${X3_CFGS:-"256:2048 256:512 256:0 128:2048"}; do
    c=${cfg%%:*}; t=${cfg##*:}
    v=$(PDF_X3_MINC=$c PDF_X3_MINT=$t timeout 300 $b 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['median_step_ms'])")
    echo "round $r  PDF_X3_MINC=$c PDF_X3_MINT=$t  img/s, ms/step, median ms: $v"
  done
done
