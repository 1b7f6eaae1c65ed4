#!/bin/bash
# weight-gradient partial tiles added with global_atomic_add_f32 instead of slabs + reduce (PDF_WG_ATOMIC=<min splits>), re-measured in round 5
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/r05_wg_atomic.txt
: > $out
B="--no-cpu-baseline --no-roofline --no-mpjpe --no-bf16-legs --no-collective-path --steps 40 --warmup 10"
run() { echo "== $*" >> $out; env "${@:2}" timeout 300 python3 $root/bench.py $B $1 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('   %.1f img/s  %.2f ms/step  median %.2f  loss %s' % (d['value'], d['ms_per_step'], d.get('median_step_ms', 0), d['config'].get('final_loss')))" >> $out; }
for v in 0 2 4; do
run "" PDF_WG_ATOMIC=$v
run "--dtype bf16 --batch 32" PDF_WG_ATOMIC=$v
done
cat $out
