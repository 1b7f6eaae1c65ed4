#!/bin/bash
# Which logical streams share a hardware queue: skip k slots of torch's stream pool before the first side stream.  -> gpurun_out/r05_stream_burn.txt
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/r05_stream_burn.txt
: > $out
B="--no-cpu-baseline --no-roofline --no-mpjpe --no-bf16-legs --no-collective-path --steps 30 --warmup 10"
run() { echo "== $*" >> $out; env "${@:2}" timeout 300 python3 $root/bench.py $B $1 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('   %.1f img/s  %.2f ms/step  median %.2f' % (d['value'], d['ms_per_step'], d.get('median_step_ms', 0)))" >> $out; }
for k in 0 1 2 3 4 5; do
run "" PDFNET_STREAM_BURN=$k PDFNET_TRUNK_GRAPH=0
done
for k in 0 1 2 3; do
run "" PDFNET_STREAM_BURN=$k PDFNET_TRUNK_GRAPH=1
done
cat $out
