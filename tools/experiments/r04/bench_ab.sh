#!/bin/bash
# usage: bench_ab.sh "<ENV=val ...>" ["<ENV=val ...>" ...] : one fp32 B=32 bench run per environment, step time + the 64x64-family symbols
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
i=0
for e in "$@"; do
  i=$((i+1))
  echo "=== [$e]"
  env $e timeout 600 python bench.py --no-cpu-baseline --no-bf16-legs --no-mpjpe --steps ${STEPS:-12} --warmup 4 2>/dev/null | tail -1 > gpurun_out/ab_$i.json
  python - "$e" gpurun_out/ab_$i.json <<'PY'
import json, sys
d = json.load(open(sys.argv[2]))
r = d['roofline']
print("%-40s %.1f img/s  %.2f ms/step   all-gemm %.1f TF (%.2f ms)  step frac %.4f" % (sys.argv[1], d['value'], d['ms_per_step'], r['all_gemm_kernels']['achieved'], r['all_gemm_kernels']['gemm_ms_per_step'], r['step_level']['frac_of_mfma_peak']))
for k, v in r['per_symbol'].items():
    if v['ms'] >= 0.3:
        print("    %-64s n=%3d %7.3f ms %6.1f TF" % (k[:64], v['launches'], v['ms'], v['tflops']))
PY
done
