#!/bin/bash
# Grouped weight gradients (functional.WGRAD_GROUP): eager fp32 / bf16 steps at group sizes 1, 4, 8, and the whole-step hipGraph with the
# weight-gradient stream forked inside the capture one launch at a time (group 1), in groups, and not at all.  -> gpurun_out/r05_wgrad_group.txt
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/r05_wgrad_group.txt
: > $out
B="--no-cpu-baseline --no-roofline --no-mpjpe --no-bf16-legs --no-collective-path --steps 30 --warmup 10"
run() { echo "== $*" >> $out; env "${@:2}" timeout 300 python3 $root/bench.py $B $1 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('   %.1f img/s  %.2f ms/step  median %.2f' % (d['value'], d['ms_per_step'], d.get('median_step_ms', 0)))" >> $out; }
for g in 1 4 8; do run "" PDFNET_WGRAD_GROUP=$g; done
for g in 1 4 8; do run "--dtype bf16 --batch 32" PDFNET_WGRAD_GROUP=$g; done
run "--graph" X=1
run "--graph" PDF_GRAPH_WGRAD_STREAM=1 PDFNET_GRAPH_WGRAD_GROUP=1
run "--graph" PDF_GRAPH_WGRAD_STREAM=1 PDFNET_GRAPH_WGRAD_GROUP=16
run "--graph" PDF_GRAPH_WGRAD_STREAM=1 PDFNET_GRAPH_WGRAD_GROUP=64
run "--graph --dtype bf16 --batch 32" X=1
run "--graph --dtype bf16 --batch 32" PDF_GRAPH_WGRAD_STREAM=1 PDFNET_GRAPH_WGRAD_GROUP=1
run "--graph --dtype bf16 --batch 32" PDF_GRAPH_WGRAD_STREAM=1 PDFNET_GRAPH_WGRAD_GROUP=16
run "--graph --dtype bf16 --batch 32" PDF_GRAPH_WGRAD_STREAM=1 PDFNET_GRAPH_WGRAD_GROUP=64
cat $out
