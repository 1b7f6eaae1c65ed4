#!/bin/bash
# attention-backward work sharing at B = 64 per GPU (128 workgroups per pass already): PDF_MESH_ATT_PARTS 224 (the B = 32 optimum) vs 112 / 122 / 124
root=${GRAFT_REPO_ROOT:-$PWD}
B="--no-cpu-baseline --no-roofline --no-mpjpe --no-bf16-legs --no-collective-path --steps 40 --warmup 10 --dtype bf16 --batch 64"
for p in 224 112 122 124 111; do
echo "== PDF_MESH_ATT_PARTS=$p"; PDF_MESH_ATT_PARTS=$p timeout 300 python3 $root/bench.py $B 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('   %.1f img/s  %.2f ms/step  median %.2f' % (d['value'], d['ms_per_step'], d.get('median_step_ms', 0)))"
done
