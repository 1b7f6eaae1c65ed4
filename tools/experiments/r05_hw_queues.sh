#!/bin/bash
# HIP streams are multiplexed onto GPU_MAX_HW_QUEUES hardware queues (ROCclr default 4); the step uses ~8 streams.  Eager and trunk-graph
# steps at 2 / 4 / 8 / 16 hardware queues.  -> gpurun_out/r05_hw_queues.txt
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/r05_hw_queues.txt
: > $out
B="--no-cpu-baseline --no-roofline --no-mpjpe --no-bf16-legs --no-collective-path --steps 30 --warmup 10"
run() { echo "== $*" >> $out; env "${@:2}" timeout 300 python3 $root/bench.py $B $1 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('   %.1f img/s  %.2f ms/step  median %.2f' % (d['value'], d['ms_per_step'], d.get('median_step_ms', 0)))" >> $out; }
for q in 2 4 8 16; do
run "" GPU_MAX_HW_QUEUES=$q PDFNET_TRUNK_GRAPH=0
run "" GPU_MAX_HW_QUEUES=$q PDFNET_TRUNK_GRAPH=1
done
for q in 4 8 16; do
run "--dtype bf16 --batch 32" GPU_MAX_HW_QUEUES=$q PDFNET_TRUNK_GRAPH=0
run "--dtype bf16 --batch 32" GPU_MAX_HW_QUEUES=$q PDFNET_TRUNK_GRAPH=1
done
cat $out
