#!/bin/bash
# igemm_nt tile order: groups of gm row-tiles (PDF_IG_GROUPM = 1 (rows of tiles one after the other), 2, 4, 8): per layer and the step; PMC fetch of the
# transposed-convolution launch that re-reads its weight panel -> gpurun_out/r05_groupm.txt
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/r05_groupm.txt
: > $out
B="--no-cpu-baseline --no-roofline --no-mpjpe --no-bf16-legs --no-collective-path --steps 40 --warmup 10"
for gm in 1 2 4 8; do
  echo "==== PDF_IG_GROUPM=$gm" >> $out
  for f in l1.conv1 l1.conv3 l2.conv1 l2.conv3 l3.conv1 l3.conv3 l4.conv3 head_3x3 feat_3x3 netR2.6; do
    PDF_IG_GROUPM=$gm PDF_BENCH_WINOGRAD=1 timeout 200 python3 $root/tools/gemm_bench.py $f 2>&1 | grep "fwd" | cut -c1-170 >> $out
  done
  for r in 1 2; do
  PDF_IG_GROUPM=$gm timeout 300 python3 $root/bench.py $B 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('   step: %.1f img/s  %.2f ms/step  median %.2f' % (d['value'], d['ms_per_step'], d.get('median_step_ms', 0)))" >> $out
  done
done
cat $out
