#!/bin/bash
# HIP streams are multiplexed onto GPU_MAX_HW_QUEUES hardware queues (ROCclr default 4); the step uses ~10 streams.  The eager step at 2..8 / 16
# hardware queues, fp32 B=32 and bf16 B=32.  -> gpurun_out/r05_hw_queues.txt   (round 5 result: 4 is the most this process gets without falling
# into a 73-75 ms mode; profiles/r05_trunk_graph_experiment.txt, which also holds the rows with the graphed trunk of commit 7e00bc8)
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/r05_hw_queues.txt
: > $out
B="--no-cpu-baseline --no-roofline --no-mpjpe --no-bf16-legs --no-collective-path --steps 30 --warmup 10"
run() { echo "== $*" >> $out; env "${@:2}" timeout 300 python3 $root/bench.py $B $1 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('   %.1f img/s  %.2f ms/step  median %.2f' % (d['value'], d['ms_per_step'], d.get('median_step_ms', 0)))" >> $out; }
for q in 2 3 4 5 6 7 8 16; do
run "" GPU_MAX_HW_QUEUES=$q
run "--dtype bf16 --batch 32" GPU_MAX_HW_QUEUES=$q
done
cat $out
