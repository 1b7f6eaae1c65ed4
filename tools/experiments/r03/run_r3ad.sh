cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3ad
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_bf16_gpu.py -x -q 2>&1 | tail -3
for u in 0 1; do
 for f in feat p2/head; do
  echo "== U $u $f"; PDF_IG_HALO256=$u timeout 120 python tools/gemm_bench.py $f 2>&1 | grep "fwd" | sed -e 's/ GF |/ |/' | cut -c1-160
 done; done
for i in 1 2; do
PDF_IG_HALO256=0 timeout 300 python bench.py --steps 20 --warmup 8 --no-cpu-baseline --no-roofline --no-mpjpe --no-bf16-legs > gpurun_out/r3ad/b_old$i.json 2>/dev/null
timeout 300 python bench.py --steps 20 --warmup 8 --no-cpu-baseline --no-roofline --no-mpjpe --no-bf16-legs > gpurun_out/r3ad/b_new$i.json 2>/dev/null
done
for f in gpurun_out/r3ad/b_*.json; do echo $f; python -c "
import json,sys
d=json.loads(open('$f').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config']['final_loss'])"; done
