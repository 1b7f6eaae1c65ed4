cd $GRAFT_REPO_ROOT
for lib in libpdfnet_hip.so libpdfnet_hip_nodeep.so libpdfnet_hip_whatif3.so libpdfnet_hip_whatif4.so; do
 for f in l1.conv3 l1.conv1 l2.conv3 l3.conv3 l3.conv2 netR1.3 netR1.6 netR2.6; do
  echo "== LIB $lib $f"; PDFNET_HIP_LIB=$GRAFT_REPO_ROOT/pdfnet_amd/$lib timeout 120 python tools/gemm_bench.py $f 2>&1 | grep "fwd" | sed -e 's/ GF |/ |/' | cut -c1-125
 done; done
