#!/bin/bash
# Why is the bf16 B=32 leg of the default run slower than the standalone bf16 run on the same box?  Legs after: (a) nothing extra, (b) the MPJPE pass,
# (c) MPJPE + CPU oracle (the driver's default), (d) --gemm-shapes / --hbm-shapes tables.
cd $GRAFT_REPO_ROOT
for f in "--no-cpu-baseline --no-mpjpe" "--no-cpu-baseline" "" "--no-cpu-baseline --no-mpjpe --gemm-shapes /tmp/g.txt --hbm-shapes /tmp/h.txt"; do
echo "== python bench.py $f"
timeout 900 python bench.py $f 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('fp32', d['value']); print({k: (v['images_per_s'], v['launch'], v['auto_choice_ms']) for k, v in d['bf16_per_gpu'].items() if isinstance(v, dict)})"
done
