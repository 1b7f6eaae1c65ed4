cd $GRAFT_REPO_ROOT
for lib in libpdfnet_hip.so libpdfnet_hip_whatif1.so libpdfnet_hip_whatif2.so; do
 for f in l2.conv2_3x3 l3.conv l4.conv2 netR2.6 netR3.3; do
  echo "== LIB $lib $f"; PDFNET_HIP_LIB=$GRAFT_REPO_ROOT/pdfnet_amd/$lib timeout 120 python tools/gemm_bench.py $f 2>&1 | grep "fwd" | sed -e 's/ GF |/ |/' | cut -c1-160
 done; done
