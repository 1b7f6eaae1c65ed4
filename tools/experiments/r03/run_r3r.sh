cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -k "conv2d or linear" 2>&1 | tail -1
for t in 200 600 1100 2100; do
 for f in l2.conv l3.conv2 dec_3x3 p2/head l1.conv3 l1.conv1 netR2.6 head_1x1; do
  echo "== T128 $t $f"; PDF_IG_T128=$t timeout 120 python tools/gemm_bench.py $f 2>&1 | grep "fwd" | sed -e 's/ GF |/ |/' | cut -c1-125
 done; done
