#!/bin/bash
# bf16 mode: the fused mesh levels' linear products on the bf16 MFMA (csrc/meshdec_bf16.hip, PDFNET_MESH_BF16_MFMA) off / on -> gpurun_out/r05_mesh_bf16_mfma.txt
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/r05_mesh_bf16_mfma.txt
: > $out
B="--no-cpu-baseline --no-roofline --no-mpjpe --no-bf16-legs --no-collective-path --steps 40 --warmup 10"
for r in 1 2; do for a in "--dtype bf16 --batch 32" "--dtype bf16 --batch 64"; do for m in 0 1; do
echo "== $a PDFNET_MESH_BF16_MFMA=$m" >> $out; PDFNET_MESH_BF16_MFMA=$m timeout 300 python3 $root/bench.py $B $a 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('   %.1f img/s  %.2f ms/step  median %.2f  loss %s' % (d['value'], d['ms_per_step'], d.get('median_step_ms', 0), d['config'].get('final_loss')))" >> $out
done; done; done
cat $out
