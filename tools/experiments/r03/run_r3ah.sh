cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3ah
PDFNET_BF16_STORAGE=1 timeout 900 python -m pytest tests/test_bf16_gpu.py -x -q > gpurun_out/r3ah/t_storage.log 2>&1; tail -5 gpurun_out/r3ah/t_storage.log | cut -c1-250
for i in 1 2; do
PDFNET_BF16_STORAGE=1 timeout 300 python bench.py --dtype bf16 --batch 64 --steps 10 --warmup 6 --no-cpu-baseline --no-roofline --no-mpjpe > gpurun_out/r3ah/b16_st$i.json 2>gpurun_out/r3ah/err_st$i.txt
timeout 300 python bench.py --dtype bf16 --batch 64 --steps 10 --warmup 6 --no-cpu-baseline --no-roofline --no-mpjpe > gpurun_out/r3ah/b16_base$i.json 2>/dev/null
done
PDFNET_BF16_STORAGE=1 timeout 300 python bench.py --dtype bf16 --batch 32 --steps 12 --warmup 6 --no-cpu-baseline --no-roofline --no-mpjpe > gpurun_out/r3ah/b16_B32_st.json 2>/dev/null
tail -3 gpurun_out/r3ah/err_st1.txt | cut -c1-300
for f in gpurun_out/r3ah/b*.json; do echo $f; python -c "
import json,sys
d=json.loads(open('$f').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config']['final_loss'])"; done
