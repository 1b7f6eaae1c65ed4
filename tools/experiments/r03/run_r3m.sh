cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3m
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -k "conv2d or linear or deconv" 2>&1 | tail -2
for f in feat_3x3 p2/head dec_3x3 l1.conv2 l2.conv2 l3.conv2 l4.conv2 l3.conv1 l3.conv3 l4.conv3 netR1.6 netR2.6 gcn; do
  timeout 120 python tools/gemm_bench.py $f 2>&1 | grep "bwd_w" | sed -e 's/ GF.*bwd_w/ bwd_w/' | cut -c1-110
done
B="--no-cpu-baseline --no-bf16-legs --no-mpjpe --no-roofline --steps 20 --warmup 6"
python bench.py $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('model', d['value'], d['ms_per_step'])"
PDF_WG_SLOTS=768 python bench.py $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('occ3all', d['value'], d['ms_per_step'])"
python bench.py $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('model', d['value'], d['ms_per_step'])"
python bench.py --gemm-shapes gpurun_out/r3m/gemm_shapes.txt --no-cpu-baseline --no-bf16-legs --no-mpjpe 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print(r['kernel'], r['achieved'], r['frac'], r['ms_per_step'], r['traffic']); a=r['all_gemm_kernels']; print(a['achieved'], a['gemm_ms_per_step'])
for k,v in list(r['per_symbol'].items())[:8]: print(' ',k,v)"
grep "bwd_weight" gpurun_out/r3m/gemm_shapes.txt | head -24 | cut -c1-150
