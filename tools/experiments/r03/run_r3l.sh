cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3l
timeout 1700 python -m pytest tests/ -x -q -m gpu 2>&1 | grep -E "passed|failed|Error" | tail -5 > gpurun_out/r3l/t_all.log
cat gpurun_out/r3l/t_all.log
python bench.py > gpurun_out/r03_bench_B32_1gpu.json 2> gpurun_out/r3l/bench.err
python bench.py --gemm-shapes gpurun_out/r03_gemm_shapes.txt --no-cpu-baseline --no-bf16-legs --no-mpjpe > /dev/null 2>&1
B="--no-cpu-baseline --no-bf16-legs --no-mpjpe --steps 20 --warmup 6"
python bench.py $B --dtype bf16 --batch 64 > gpurun_out/r03_bench_bf16_B64_1gpu.json 2>/dev/null
python bench.py $B --dtype bf16 --batch 32 > gpurun_out/r03_bench_bf16_B32_1gpu.json 2>/dev/null
python bench.py $B --config rgb-encoder > gpurun_out/r03_bench_rgb_encoder_B8.json 2>/dev/null
python bench.py $B --batch 8 --no-roofline > gpurun_out/r03_bench_B8_1gpu.json 2>/dev/null
python bench.py $B --graph --no-roofline > gpurun_out/r03_bench_B32_graph.json 2>/dev/null
for f in gpurun_out/r03_bench_*.json; do echo $f; python -c "import json,sys; d=json.load(open('$f')); print(d['value'], d['ms_per_step'])"; done
python __graft_entry__.py --smoke 2>&1 | tail -2
