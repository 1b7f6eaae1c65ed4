cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_trainer_gpu.py tests/test_configs_gpu.py -x -q 2>&1 | grep -E "passed|failed|Error" | tail -3
B="--no-cpu-baseline --no-bf16-legs --no-mpjpe --no-roofline --steps 20 --warmup 6"
for i in 1 2; do
PDFNET_OVERLAP_ADAM=0 python bench.py $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('no-overlap', d['value'], d['ms_per_step'])"
python bench.py $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('overlap   ', d['value'], d['ms_per_step'])"
done
