cd $GRAFT_REPO_ROOT
for v in 0 1; do for f in netR l1.conv3 l1.conv1 l2.conv3; do
  echo "== BK32PLAIN=$v $f"; PDF_IG_BK32_PLAIN=$v timeout 120 python tools/gemm_bench.py $f 2>&1 | grep "fwd" | cut -c1-150
done; done
B="--no-cpu-baseline --no-bf16-legs --no-mpjpe --no-roofline --steps 20 --warmup 6"
for i in 1 2; do
PDF_IG_BK32_PLAIN=0 python bench.py $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bk16', d['value'], d['ms_per_step'])"
PDF_IG_BK32_PLAIN=1 python bench.py $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bk32', d['value'], d['ms_per_step'])"
done
