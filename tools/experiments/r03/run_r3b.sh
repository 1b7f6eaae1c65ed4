mkdir -p gpurun_out/r3b; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "statistics or batchnorm or conv2d" 2>&1 | tail -8 > gpurun_out/r3b/t_ops.log
timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_full_gradient_gpu.py tests/test_trainer_gpu.py tests/test_configs_gpu.py tests/test_modules_gpu.py -q 2>&1 | tail -30 > gpurun_out/r3b/t_model.log
B="--no-cpu-baseline --no-bf16-legs --no-mpjpe --steps 20 --warmup 5"
PDFNET_BN_EPILOGUE_STATS=0 python bench.py $B > gpurun_out/r3b/b_nostats.json 2>/dev/null
python bench.py $B > gpurun_out/r3b/b_stats.json 2>/dev/null
PDFNET_BN_EPILOGUE_STATS=0 python bench.py $B --no-roofline > gpurun_out/r3b/b_nostats2.json 2>/dev/null
python bench.py $B --no-roofline > gpurun_out/r3b/b_stats2.json 2>/dev/null
python bench.py $B --no-roofline --graph > gpurun_out/r3b/b_graph.json 2>/dev/null
PDF_GRAPH_WGRAD_STREAM=1 python bench.py $B --no-roofline --graph > gpurun_out/r3b/b_graph_wg.json 2>gpurun_out/r3b/b_graph_wg.err
python bench.py $B --no-roofline --dtype bf16 --batch 64 > gpurun_out/r3b/b_bf16_64.json 2>/dev/null
python bench.py $B --no-roofline --dtype bf16 --batch 32 > gpurun_out/r3b/b_bf16_32.json 2>/dev/null
cat gpurun_out/r3b/t_ops.log gpurun_out/r3b/t_model.log
for f in gpurun_out/r3b/b_*.json; do echo $f; python -c "import json,sys; d=json.load(open('$f')); print(d['value'], d['ms_per_step'])"; done
