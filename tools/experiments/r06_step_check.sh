# the x3 tile rule in the step, alternating within one call (boxes differ by a few per cent: only same-call comparisons count)
cd ${GRAFT_REPO_ROOT:-$PWD}
b="python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-mpjpe --no-bf16-legs --no-roofline --no-native-leg --no-collective-path"
for r in 1 2; do
  for rule in 0 2 1; do
    v=$(PDF_X3_TILE_RULE=$rule timeout 300 $b 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['median_step_ms'])")
    echo "round $r  PDF_X3_TILE_RULE=$rule  img/s, ms/step, median ms: $v"
  done
done
