cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -k "conv2d or deconv or linear" 2>&1 | tail -1
for lib in libpdfnet_hip_head.so libpdfnet_hip.so libpdfnet_hip_store7.so; do
 for f in l1.conv3 l2.conv l3.conv l4.conv netR1.3 netR2.6 netR3.3; do
  echo "== LIB $lib $f"; PDFNET_HIP_LIB=$GRAFT_REPO_ROOT/pdfnet_amd/$lib timeout 120 python tools/gemm_bench.py $f 2>&1 | grep "fwd" | sed -e 's/ GF |/ |/' | cut -c1-160
 done; done
