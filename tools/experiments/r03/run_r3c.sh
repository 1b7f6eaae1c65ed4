cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3c
tools/probe/buf_lds_probe > gpurun_out/r3c/probe.txt 2>&1
timeout 900 python -m pytest tests/test_bf16_gpu.py -x -q 2>&1 | tail -8 > gpurun_out/r3c/t_bf16.log
timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_trainer_gpu.py -x -q -k "statistics or graph_replay or evaluation_writes" 2>&1 | tail -8 > gpurun_out/r3c/t_misc.log
B="--no-cpu-baseline --no-bf16-legs --no-mpjpe --no-roofline --steps 20 --warmup 6"
python bench.py $B --dtype bf16 --batch 64 > gpurun_out/r3c/b_bf16_64.json 2>/dev/null
python bench.py $B --dtype bf16 --batch 32 > gpurun_out/r3c/b_bf16_32.json 2>/dev/null
cat gpurun_out/r3c/probe.txt gpurun_out/r3c/t_bf16.log gpurun_out/r3c/t_misc.log
for f in gpurun_out/r3c/b_*.json; do echo $f; python -c "import json,sys; d=json.load(open('$f')); print(d['value'], d['ms_per_step'])"; done
