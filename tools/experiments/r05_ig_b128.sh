#!/bin/bash
# igemm_nt operand fragments as ds_read_b128 (PDF_IG_B128, compile time): .ab/libpdf_b128_0.so = the [BK + 1] image with scalar LDS accesses,
# pdfnet_amd/libpdfnet_hip.so = the b128 form.  Per-layer (tools/gemm_bench.py) and the step.  -> gpurun_out/r05_ig_b128.txt
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/r05_ig_b128.txt
: > $out
B="--no-cpu-baseline --no-roofline --no-mpjpe --no-bf16-legs --no-collective-path --steps 40 --warmup 10"
for lib in .ab/libpdf_b128_0.so pdfnet_amd/libpdfnet_hip.so; do
  echo "==== $lib" >> $out
  for f in l1.conv1 l1.conv2 l1.conv3 l2.conv2_3x3 l2.conv1 l2.conv3 l3.conv2 l3.conv1 l3.conv3 l4.conv2 l4.conv3 head_3x3 feat_3x3 netR1.3 netR2.6; do
    PDFNET_HIP_LIB=$root/$lib PDF_BENCH_WINOGRAD=1 timeout 200 python3 $root/tools/gemm_bench.py $f 2>&1 | grep "fwd" | cut -c1-170 >> $out
  done
  for r in 1 2; do
  PDFNET_HIP_LIB=$root/$lib timeout 300 python3 $root/bench.py $B 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('   step: %.1f img/s  %.2f ms/step  median %.2f  loss %s' % (d['value'], d['ms_per_step'], d.get('median_step_ms', 0), d['config'].get('final_loss')))" >> $out
  done
done
cat $out
