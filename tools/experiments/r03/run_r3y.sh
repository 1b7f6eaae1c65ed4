cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3y
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_modules_gpu.py -x -q 2>&1 | grep -E "passed|failed" | tail -1
for i in 1 2; do
PDFNET_HIP_LIB=$GRAFT_REPO_ROOT/pdfnet_amd/libpdfnet_hip_nodeep.so timeout 300 python bench.py --steps 20 --warmup 8 --no-cpu-baseline --no-roofline --no-mpjpe --no-bf16-legs > gpurun_out/r3y/b_old$i.json 2>/dev/null
timeout 300 python bench.py --steps 20 --warmup 8 --no-cpu-baseline --no-roofline --no-mpjpe --no-bf16-legs > gpurun_out/r3y/b_new$i.json 2>/dev/null
done
for f in gpurun_out/r3y/b_*.json; do echo $f; python -c "
import json,sys
d=json.loads(open('$f').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; done
