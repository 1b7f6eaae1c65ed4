#!/bin/bash
# mesh attention backward: the dq and the dk / dv pass in separate workgroups (PDF_MESH_ATT_SPLIT, bit per level) and their (head, tile) items shared by
# PDF_MESH_ATT_PARTS workgroups each (digits: level 0 / 1 / 2)
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/r05_att_split.txt
: > $out
B="--no-cpu-baseline --no-roofline --no-mpjpe --no-bf16-legs --no-collective-path --steps 40 --warmup 10"
run() { echo "==== $*" >> $out; for r in 1 2; do env "$@" timeout 300 python3 $root/bench.py $B 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('   step: %.1f img/s  %.2f ms/step  median %.2f' % (d['value'], d['ms_per_step'], d.get('median_step_ms', 0)))" >> $out; done; }
run PDF_MESH_ATT_SPLIT=0
run PDF_MESH_ATT_PARTS=111
run PDF_MESH_ATT_PARTS=122
run PDF_MESH_ATT_PARTS=124
run PDF_MESH_ATT_PARTS=224
run PDF_MESH_ATT_PARTS=248
cat $out
