#!/bin/bash
# The ResNet trunk replayed as two hipGraphs inside the eager step (pdfnet_amd/graphed.py): fp32 B=32, bf16 B=32 / B=64, off / on, and the
# weight-gradient group size inside the backward graph.  -> gpurun_out/r05_trunk_graph.txt
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/r05_trunk_graph.txt
: > $out
B="--no-cpu-baseline --no-roofline --no-mpjpe --no-bf16-legs --no-collective-path --steps 30 --warmup 10"
run() { echo "== $*" >> $out; env "${@:2}" timeout 300 python3 $root/bench.py $B $1 2>$root/gpurun_out/r05_trunk_graph.err | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('   %.1f img/s  %.2f ms/step  median %.2f  loss %s' % (d['value'], d['ms_per_step'], d.get('median_step_ms', 0), d['config'].get('final_loss')))" >> $out; tail -3 $root/gpurun_out/r05_trunk_graph.err | grep -i "error\|Traceback" >> $out; }
for a in "" "--dtype bf16 --batch 32" "--dtype bf16 --batch 64"; do
run "$a" PDFNET_TRUNK_GRAPH=0
run "$a" PDFNET_TRUNK_GRAPH=1
done
cat $out
