#!/bin/bash
# trunk graph on the default (null) stream vs on a created stream
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/r05_trunk_graph2.txt
: > $out
B="--no-cpu-baseline --no-roofline --no-mpjpe --no-bf16-legs --no-collective-path --steps 30 --warmup 10"
run() { echo "== $*" >> $out; env "${@:2}" timeout 300 python3 $root/bench.py $B $1 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('   %.1f img/s  %.2f ms/step  median %.2f' % (d['value'], d['ms_per_step'], d.get('median_step_ms', 0)))" >> $out; }
for st in 2; do for g in 0 1; do
run "" PDFNET_BENCH_STREAM=$st PDFNET_TRUNK_GRAPH=$g
run "--dtype bf16 --batch 32" PDFNET_BENCH_STREAM=$st PDFNET_TRUNK_GRAPH=$g
done; done
cat $out
