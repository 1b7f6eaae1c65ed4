cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3h
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -k "gather_sub or group_then" 2>&1 | tail -4 > gpurun_out/r3h/t_ops.log
bash tools/probe/sq_bf16.sh > gpurun_out/r3h/sq_bf16.txt 2>&1
B="--no-cpu-baseline --no-bf16-legs --no-mpjpe --no-roofline --steps 20 --warmup 6"
PDFNET_GATHER_SORTED=0 python bench.py $B > gpurun_out/r3h/b_atomic.json 2>/dev/null
python bench.py $B > gpurun_out/r3h/b_sorted.json 2>/dev/null
cat gpurun_out/r3h/t_ops.log gpurun_out/r3h/sq_bf16.txt
for f in gpurun_out/r3h/b_*.json; do echo $f; python -c "import json,sys; d=json.load(open('$f')); print(d['value'], d['ms_per_step'])"; done
