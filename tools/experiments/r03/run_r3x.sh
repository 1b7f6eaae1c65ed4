cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q 2>&1 | tail -1
for lib in libpdfnet_hip.so libpdfnet_hip_lean8192.so libpdfnet_hip_lean16384.so; do
 for f in l1.conv l2.conv1 netR1 netR2.0 dec_ head_ p2/head feat; do
  echo "== LIB $lib $f"; PDFNET_HIP_LIB=$GRAFT_REPO_ROOT/pdfnet_amd/$lib timeout 120 python tools/gemm_bench.py $f 2>&1 | grep "fwd" | sed -e 's/ GF |/ |/' | cut -c1-125
 done; done
