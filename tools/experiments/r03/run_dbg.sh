cd $GRAFT_REPO_ROOT
for i in $(seq 1 10); do timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -k "statistics_from" 2>&1 | grep -E "AssertionError|passed|failed" | cut -c1-250 | head -4; done
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q 2>&1 | tail -3
timeout 1200 python -m pytest tests/test_model_gpu.py tests/test_full_gradient_gpu.py tests/test_trainer_gpu.py -x -q 2>&1 | tail -3
for v in 0 1; do for f in feat_3x3 p2/head dec_3x3 l3.conv2 l4.conv2 l3.conv1 netR1.6; do
  echo "== UNI=$v $f"; PDF_WG_UNIFORM=$v timeout 120 python tools/gemm_bench.py $f 2>&1 | grep "bwd_w" | sed -e 's/.*bwd_w/bwd_w/' | cut -c1-60
done; done
B="--no-cpu-baseline --no-bf16-legs --no-mpjpe --no-roofline --steps 20 --warmup 6"
PDF_WG_UNIFORM=0 python bench.py $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('uni0', d['value'], d['ms_per_step'])"
python bench.py $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('uni1', d['value'], d['ms_per_step'])"
PDF_WG_UNIFORM=0 python bench.py $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('uni0', d['value'], d['ms_per_step'])"
python bench.py $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('uni1', d['value'], d['ms_per_step'])"
