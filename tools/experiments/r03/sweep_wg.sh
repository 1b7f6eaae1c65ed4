# weight-gradient split sweep (launch_wgemm: PDF_WG_TARGET = target block count, PDF_WG_QUANT=0: no adjustment)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/sweep
for f in feat_3x3 p2/head dec_3x3 l3.conv2 l2.conv2_3x3 l4.conv2 l3.conv1 l1.conv2; do
  for t in 384 576 720 768 1008 1152 1536 2304 3072; do
    echo "== $f target $t"; PDF_WG_QUANT=0 PDF_WG_TARGET=$t timeout 120 python tools/gemm_bench.py $f 2>&1 | grep "bwd_w" | sed -e 's/.*bwd_data/bwd_data/' | cut -c1-120
  done
done > gpurun_out/sweep/wg_target.txt 2>&1
for f in l4.conv2 l3.conv2 l4.conv3 dec_1x1; do
  for cfg in "128 320" "256 512" "256 768" "512 768" "512 1024"; do
    set -- $cfg; echo "== $f splitk maxt $1 target $2"; PDF_IG_SPLITK_MAXT=$1 PDF_IG_SPLITK_TARGET=$2 timeout 120 python tools/gemm_bench.py $f 2>&1 | grep "fwd" | cut -c1-150
  done
done > gpurun_out/sweep/ig_splitk.txt 2>&1
tail -n 200 gpurun_out/sweep/wg_target.txt; cat gpurun_out/sweep/ig_splitk.txt
