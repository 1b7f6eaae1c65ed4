cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3z
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_bf16_gpu.py -x -q 2>&1 | grep -E "passed|failed|Error" | tail -3
PDF_BENCH_BF16=1 timeout 200 python tools/gemm_bench.py l1.conv 2>&1 | grep fwd | cut -c1-130
PDF_BENCH_BF16=1 timeout 200 python tools/gemm_bench.py netR1 2>&1 | grep fwd | cut -c1-130
PDF_BENCH_BF16=1 timeout 200 python tools/gemm_bench.py l3.conv 2>&1 | grep fwd | cut -c1-130
PDF_BENCH_BF16=1 timeout 200 python tools/gemm_bench.py feat 2>&1 | grep fwd | cut -c1-130
for i in 1 2; do
timeout 300 python bench.py --dtype bf16 --batch 64 --steps 10 --warmup 6 --no-cpu-baseline --no-roofline --no-mpjpe > gpurun_out/r3z/b16_new$i.json 2>/dev/null
PDF_IG_BF16_STATS=0 timeout 300 python bench.py --dtype bf16 --batch 64 --steps 10 --warmup 6 --no-cpu-baseline --no-roofline --no-mpjpe > gpurun_out/r3z/b16_nostat$i.json 2>/dev/null
PDF_IG_BF16_STATS=0 PDF_IG_BUFSTORE=0 timeout 300 python bench.py --dtype bf16 --batch 64 --steps 10 --warmup 6 --no-cpu-baseline --no-roofline --no-mpjpe > gpurun_out/r3z/b16_old$i.json 2>/dev/null
done
timeout 300 python bench.py --dtype bf16 --batch 32 --steps 12 --warmup 6 --no-cpu-baseline --no-roofline --no-mpjpe > gpurun_out/r3z/b16_B32.json 2>/dev/null
for f in gpurun_out/r3z/b*.json; do echo $f; python -c "
import json,sys
d=json.loads(open('$f').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; done
