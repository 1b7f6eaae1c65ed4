cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3d
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_bf16_gpu.py -x -q 2>&1 | tail -6 > gpurun_out/r3d/t_ops.log
for v in 0 1; do for f in feat_3x3 p2/head dec_3x3 l3.conv2 l2.conv2_3x3 l3.conv1 netR1.6; do
  echo "== BUF=$v $f"; PDF_WG_BUF=$v timeout 120 python tools/gemm_bench.py $f 2>&1 | grep "bwd_w" | sed -e 's/.*bwd_data/bwd_data/' | cut -c1-120
done; done > gpurun_out/r3d/wg_buf.txt 2>&1
B="--no-cpu-baseline --no-bf16-legs --no-mpjpe --no-roofline --steps 20 --warmup 6"
PDF_WG_BUF=0 python bench.py $B > gpurun_out/r3d/b_fp32_nobuf.json 2>/dev/null
python bench.py $B > gpurun_out/r3d/b_fp32_buf.json 2>/dev/null
PDF_WG_BUF=0 python bench.py $B > gpurun_out/r3d/b_fp32_nobuf2.json 2>/dev/null
python bench.py $B > gpurun_out/r3d/b_fp32_buf2.json 2>/dev/null
python bench.py $B --dtype bf16 --batch 64 > gpurun_out/r3d/b_bf16_64.json 2>/dev/null
python bench.py $B --dtype bf16 --batch 32 > gpurun_out/r3d/b_bf16_32.json 2>/dev/null
cat gpurun_out/r3d/t_ops.log gpurun_out/r3d/wg_buf.txt
for f in gpurun_out/r3d/b_*.json; do echo $f; python -c "import json,sys; d=json.load(open('$f')); print(d['value'], d['ms_per_step'])"; done
