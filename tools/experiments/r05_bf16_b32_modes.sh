#!/bin/bash
# bf16 B=32 is GPU-bound now (r05_trunk_tape.txt): do bf16 storage of the conv -> BatchNorm tensors and epilogue statistics pay at B=32 too?
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/r05_bf16_b32_modes.txt
: > $out
B="--no-cpu-baseline --no-roofline --no-mpjpe --no-bf16-legs --no-collective-path --steps 40 --warmup 10 --dtype bf16 --batch 32"
run() { echo "== $*" >> $out; env "$@" timeout 300 python3 $root/bench.py $B 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('   %.1f img/s  %.2f ms/step  median %.2f  loss %s' % (d['value'], d['ms_per_step'], d.get('median_step_ms', 0), d['config'].get('final_loss')))" >> $out; }
for r in 1 2; do
run X=1
run PDFNET_BF16_STORAGE=1
run PDFNET_BN_EPILOGUE_STATS_BF16=1
run PDFNET_BF16_STORAGE=1 PDFNET_BN_EPILOGUE_STATS_BF16=1
done
cat $out
