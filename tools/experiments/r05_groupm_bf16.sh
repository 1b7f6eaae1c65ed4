#!/bin/bash
# tile order gm = 1 / 4 / 8 for the bf16 kernels too (they are bound by what they pull from L2): bf16 B=64 and B=32, and the fp32 step again
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/r05_groupm_bf16.txt
: > $out
B="--no-cpu-baseline --no-roofline --no-mpjpe --no-bf16-legs --no-collective-path --steps 40 --warmup 10"
run() { echo "== $*" >> $out; env "${@:2}" timeout 300 python3 $root/bench.py $B $1 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('   %.1f img/s  %.2f ms/step  median %.2f' % (d['value'], d['ms_per_step'], d.get('median_step_ms', 0)))" >> $out; }
for gm in 1 4 8; do
run "--dtype bf16 --batch 64" PDF_IG_GROUPM=$gm
run "--dtype bf16 --batch 32" PDF_IG_GROUPM=$gm
run "" PDF_IG_GROUPM=$gm
done
cat $out
