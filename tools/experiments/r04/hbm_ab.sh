#!/bin/bash
# usage: hbm_ab.sh "<ENV=val ...>" ... : fp32 B=32 bench per environment: step time + the HBM-bound entry points
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
i=0
for e in "$@"; do
  i=$((i+1))
  env $e timeout 600 python bench.py --no-cpu-baseline --no-bf16-legs --no-mpjpe --steps ${STEPS:-12} --warmup 4 2>/dev/null | tail -1 > gpurun_out/hb_$i.json
  python - "$e" gpurun_out/hb_$i.json <<'PY'
import json, sys
d = json.load(open(sys.argv[2]))
h = d['roofline_hbm']
print("%-40s %.1f img/s  %.2f ms/step   hbm-family %.0f GB/s (%.2f ms)" % (sys.argv[1], d['value'], d['ms_per_step'], h['all_hbm_bound_entry_points']['achieved'], h['all_hbm_bound_entry_points']['ms_per_step']))
for k, v in h['per_entry_point'].items():
    if v['ms'] >= 0.15:
        print("    %-30s n=%3d %7.3f ms %7.0f GB/s" % (k, v['calls'], v['ms'], v['GBs']))
PY
done
