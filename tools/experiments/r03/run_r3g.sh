cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3g
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -k "conv2d or linear or deconv" 2>&1 | tail -4 > gpurun_out/r3g/t_ops.log
for lib in libpdfnet_hip_nopipe.so libpdfnet_hip.so; do for f in feat_3x3 p2/head dec_3x3 l3.conv2 l2.conv2_3x3 l4.conv2 l3.conv1; do
  echo "== $lib $f"; PDFNET_HIP_LIB=$GRAFT_REPO_ROOT/pdfnet_amd/$lib timeout 120 python tools/gemm_bench.py $f 2>&1 | grep "fwd" | cut -c1-150
done; done > gpurun_out/r3g/pipe.txt 2>&1
B="--no-cpu-baseline --no-bf16-legs --no-mpjpe --no-roofline --steps 20 --warmup 6"
PDFNET_HIP_LIB=$GRAFT_REPO_ROOT/pdfnet_amd/libpdfnet_hip_nopipe.so python bench.py $B > gpurun_out/r3g/b_nopipe.json 2>/dev/null
python bench.py $B > gpurun_out/r3g/b_pipe.json 2>/dev/null
PDFNET_HIP_LIB=$GRAFT_REPO_ROOT/pdfnet_amd/libpdfnet_hip_nopipe.so python bench.py $B > gpurun_out/r3g/b_nopipe2.json 2>/dev/null
python bench.py $B > gpurun_out/r3g/b_pipe2.json 2>/dev/null
python bench.py --no-cpu-baseline --no-mpjpe > gpurun_out/r3g/b_full.json 2>/dev/null
cat gpurun_out/r3g/t_ops.log
python - <<'PY'
import re
rows={}
cur=None
for l in open('gpurun_out/r3g/pipe.txt'):
    m=re.match(r'== (\S+) (\S+)', l)
    if m: cur='0' if 'nopipe' in m.group(1) else '1'; continue
    m=re.match(r'(\S+)\s+M=.*?fwd\s+([\d.]+) ms\s+([\d.]+) TF.*?bwd_data\s+([\d.]+) ms\s+([\d.]+) TF.*?bwd_w\s+([\d.]+) ms\s+([\d.]+) TF', l)
    if m: rows.setdefault(m.group(1),{})[cur]=(float(m.group(3)), float(m.group(5)), float(m.group(7)))
for k,v in rows.items():
    if '0' in v and '1' in v: print('%-18s fwd %6.1f -> %6.1f   bwd_data %6.1f -> %6.1f  bwd_w %6.1f -> %6.1f' % (k, v['0'][0], v['1'][0], v['0'][1], v['1'][1], v['0'][2], v['1'][2]))
PY
for f in gpurun_out/r3g/b_*.json; do echo $f; python -c "import json,sys; d=json.load(open('$f')); print(d['value'], d['ms_per_step'])"; done
python -c "
import json; d=json.load(open('gpurun_out/r3g/b_full.json'))
print(json.dumps(d.get('bf16_per_gpu'), indent=1))
r=d['roofline']; print(r['kernel'], r['achieved'], r['frac'], r['launches_per_step'], r['ms_per_step'])
for k,v in list(r['per_symbol'].items())[:14]: print(k, v)
print(r['all_gemm_kernels']['achieved'], r['all_gemm_kernels']['gemm_ms_per_step'])
"
