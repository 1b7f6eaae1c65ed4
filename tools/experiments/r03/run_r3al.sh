cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_full_gradient_gpu.py tests/test_modules_gpu.py -x -q > gpurun_out/t.log 2>&1; grep -E "passed|failed|Error" gpurun_out/t.log | tail -4
for i in 1 2; do
PDFNET_LAZY_SA_BN=0 timeout 300 python bench.py --steps 20 --warmup 8 --no-cpu-baseline --no-roofline --no-mpjpe --no-bf16-legs 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lazy=0', d['value'], d['ms_per_step'], d['config']['final_loss'])"
timeout 300 python bench.py --steps 20 --warmup 8 --no-cpu-baseline --no-roofline --no-mpjpe --no-bf16-legs 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lazy=1', d['value'], d['ms_per_step'], d['config']['final_loss'])"
done
