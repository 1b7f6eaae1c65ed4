# the B=32 fp32 step with x3 off / on for the wide layers only (PDF_X3_MINC=512: `feat`) / on for every F(4x4) layer (MINC=128), alternating, 2 rounds
cd ${GRAFT_REPO_ROOT:-$PWD}
b="python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-mpjpe --no-bf16-legs --no-roofline --no-collective-path"
for r in 1 2; do
  for cfg in "0 512" "1 512" "1 128"; do
    set -- $cfg
    v=$(PDF_X3=$1 PDF_X3_MINC=$2 timeout 300 $b 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['median_step_ms'])")
    echo "round $r  PDF_X3=$1 PDF_X3_MINC=$2  img/s, ms/step, median ms: $v"
  done
done
