#!/bin/bash
# heads on x0 share one Winograd input transform (F.share_winograd_input): off (PDFNET_WINOGRAD_SHARE=0) / on -> gpurun_out/r05_wino_share.txt
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/r05_wino_share.txt
: > $out
B="--no-cpu-baseline --no-roofline --no-mpjpe --no-bf16-legs --no-collective-path --steps 40 --warmup 10"
run() { echo "== $*" >> $out; env "${@:2}" timeout 300 python3 $root/bench.py $B $1 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('   %.1f img/s  %.2f ms/step  median %.2f  loss %s' % (d['value'], d['ms_per_step'], d.get('median_step_ms', 0), d['config'].get('final_loss')))" >> $out; }
for r in 1 2; do for l in 0 1; do
run "" PDFNET_WINOGRAD_SHARE=$l
done; done
cat $out
