#!/bin/bash
# bf16 mode: fused mesh decoder (fp32 math on 64-128 CUs) vs the per-op bf16 chain (full-chip bf16 GEMMs, ~330 launches per level and direction)
root=${GRAFT_REPO_ROOT:-$PWD}
B="--no-cpu-baseline --no-roofline --no-mpjpe --no-bf16-legs --no-collective-path --steps 40 --warmup 10"
for a in "--dtype bf16 --batch 32" "--dtype bf16 --batch 64"; do for m in 1 0; do
echo "== $a PDFNET_MESH_FUSED_BF16=$m"; PDFNET_MESH_FUSED_BF16=$m timeout 300 python3 $root/bench.py $B $a 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('   %.1f img/s  %.2f ms/step' % (d['value'], d['ms_per_step']))"
done; done
