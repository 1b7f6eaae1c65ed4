cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3e
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q 2>&1 | tail -6 > gpurun_out/r3e/t_ops.log
for v in 0 1; do for f in l1.conv l2.conv l3.conv l4.conv p2/head dec_3x3 dec_1x1 head_1x1 netR gcn attn; do
  echo "== IGBUF=$v $f"; PDF_IG_BUF=$v timeout 120 python tools/gemm_bench.py $f 2>&1 | grep "fwd" | cut -c1-150
done; done > gpurun_out/r3e/ig_buf.txt 2>&1
B="--no-cpu-baseline --no-bf16-legs --no-mpjpe --no-roofline --steps 20 --warmup 6"
PDF_IG_BUF=0 python bench.py $B > gpurun_out/r3e/b_fp32_nobuf.json 2>/dev/null
python bench.py $B > gpurun_out/r3e/b_fp32_buf.json 2>/dev/null
PDF_IG_BUF=0 python bench.py $B > gpurun_out/r3e/b_fp32_nobuf2.json 2>/dev/null
python bench.py $B > gpurun_out/r3e/b_fp32_buf2.json 2>/dev/null
python bench.py $B --config rgb-encoder > gpurun_out/r3e/b_rgb.json 2>/dev/null
cat gpurun_out/r3e/t_ops.log
python - <<'PY'
import re
rows={}
cur=None
for l in open('gpurun_out/r3e/ig_buf.txt'):
    m=re.match(r'== IGBUF=(\d) (\S+)', l)
    if m: cur=m.group(1); continue
    m=re.match(r'(\S+)\s+M=.*?fwd\s+([\d.]+) ms\s+([\d.]+) TF.*?bwd_data\s+([\d.]+) ms\s+([\d.]+) TF', l)
    if m: rows.setdefault(m.group(1),{})[cur]=(float(m.group(3)), float(m.group(5)))
for k,v in rows.items():
    if '0' in v and '1' in v: print('%-18s fwd %6.1f -> %6.1f   bwd_data %6.1f -> %6.1f' % (k, v['0'][0], v['1'][0], v['0'][1], v['1'][1]))
PY
for f in gpurun_out/r3e/b_*.json; do echo $f; python -c "import json,sys; d=json.load(open('$f')); print(d['value'], d['ms_per_step'])"; done
