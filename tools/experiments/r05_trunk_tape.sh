#!/bin/bash
# The ResNet trunk replayed from a tape of its library calls (pdfnet_amd/taped.py, PDFNET_TRUNK_TAPE) off / on -> gpurun_out/r05_trunk_tape.txt
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/r05_trunk_tape.txt
: > $out
B="--no-cpu-baseline --no-roofline --no-mpjpe --no-bf16-legs --no-collective-path --steps 40 --warmup 10"
run() { echo "== $*" >> $out; env "${@:2}" timeout 300 python3 $root/bench.py $B $1 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('   %.1f img/s  %.2f ms/step  median %.2f  loss %s' % (d['value'], d['ms_per_step'], d.get('median_step_ms', 0), d['config'].get('final_loss')))" >> $out; }
for a in "" "--dtype bf16 --batch 32" "--dtype bf16 --batch 64" "--batch 8"; do
for t in 0 1; do
run "$a" PDFNET_TRUNK_TAPE=$t
done; done
for t in 0 1; do echo "== host time, PDFNET_TRUNK_TAPE=$t" >> $out; PDFNET_TRUNK_TAPE=$t timeout 250 python3 $root/tools/host_time.py bf16 32 2>&1 | grep -v "^W\|amdgpu.ids" | tail -3 >> $out; done
cat $out
