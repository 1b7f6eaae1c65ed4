cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -k "conv2d or linear" 2>&1 | tail -1
for lib in libpdfnet_hip_nodeep.so libpdfnet_hip.so; do
 for f in l1.conv l2.conv l3.conv l4.conv netR dec_ head_1x1; do
  echo "== LIB $lib $f"; PDFNET_HIP_LIB=$GRAFT_REPO_ROOT/pdfnet_amd/$lib timeout 120 python tools/gemm_bench.py $f 2>&1 | grep "fwd" | sed -e 's/ GF |/ |/' | cut -c1-125
 done; done
