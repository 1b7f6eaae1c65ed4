cd $GRAFT_REPO_ROOT
for lib in libpdfnet_hip.so libpdfnet_hip_whatif1.so libpdfnet_hip_whatif2.so; do
 for f in l1.conv netR1.3 netR1.6 l3.conv1 l3.conv3 l4.conv3; do
  echo "== LIB $lib $f"; PDFNET_HIP_LIB=$GRAFT_REPO_ROOT/pdfnet_amd/$lib timeout 120 python tools/gemm_bench.py $f 2>&1 | grep "fwd" | sed -e 's/ GF |/ |/' | cut -c1-160
 done; done
