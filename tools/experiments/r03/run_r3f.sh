cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3f
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_bf16_gpu.py -x -q 2>&1 | tail -6 > gpurun_out/r3f/t_ops.log
timeout 1200 python -m pytest tests/test_model_gpu.py tests/test_full_gradient_gpu.py tests/test_configs_gpu.py -x -q 2>&1 | tail -6 > gpurun_out/r3f/t_model.log
for f in feat_3x3 p2/head; do python tools/gemm_bench.py $f 2>&1 | grep fwd | cut -c1-150; done > gpurun_out/r3f/halo.txt
B="--no-cpu-baseline --no-bf16-legs --no-mpjpe --no-roofline --steps 20 --warmup 6"
python bench.py $B > gpurun_out/r3f/b_fp32.json 2>/dev/null
python bench.py $B > gpurun_out/r3f/b_fp32_2.json 2>/dev/null
python bench.py $B --dtype bf16 --batch 64 > gpurun_out/r3f/b_bf16_64.json 2>/dev/null
python bench.py $B --dtype bf16 --batch 32 > gpurun_out/r3f/b_bf16_32.json 2>/dev/null
python bench.py $B --batch 8 > gpurun_out/r3f/b_fp32_B8.json 2>/dev/null
cat gpurun_out/r3f/t_ops.log gpurun_out/r3f/t_model.log gpurun_out/r3f/halo.txt
for f in gpurun_out/r3f/b_*.json; do echo $f; python -c "import json,sys; d=json.load(open('$f')); print(d['value'], d['ms_per_step'])"; done
