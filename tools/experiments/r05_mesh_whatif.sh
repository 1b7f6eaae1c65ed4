#!/bin/bash
# What-if builds of the fused mesh decoder (results wrong on purpose, timing only): one part compiled out per build (-DMD_WHATIF=n, built in the
# build container into .ab/libpdf_n.so), loaded through PDFNET_HIP_LIB.
# 1 no ELL products, 2 no LayerNorm, 3 products cut to 4 K-steps, 4 no attention, 5 weight rows fetched from one row (no strided fetch), 6 no global -> LDS row loads
cd $GRAFT_REPO_ROOT
o=gpurun_out/r05_mesh_whatif.txt
: > $o
for v in 0 1 2 3 4 5 6; do
  lib=.ab/libpdf_$v.so
  [ $v = 0 ] && lib=pdfnet_amd/libpdfnet_hip.so
  echo "== MD_WHATIF=$v" >> $o
  PDFNET_HIP_LIB=$PWD/$lib timeout 300 python tools/mesh_bench.py 32 2>&1 | grep "fused " | grep -v unfused >> $o
done
cat $o
