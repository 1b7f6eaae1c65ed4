#!/bin/bash
# usage: bf16_ab.sh B "<ENV=val ...>" ... : bf16 train-step bench (bench.py --dtype bf16 --batch B) per environment
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
B=$1; shift
i=0
for e in "$@"; do
  i=$((i+1))
  env $e timeout 600 python bench.py --dtype bf16 --batch $B --no-cpu-baseline --no-mpjpe --steps ${STEPS:-14} --warmup 5 2>/dev/null | tail -1 > gpurun_out/bf_$i.json
  python - "$e" gpurun_out/bf_$i.json <<'PY'
import json, sys
d = json.load(open(sys.argv[2]))
r = d['roofline']
print("%-52s %.1f img/s  %.2f ms/step   gemm %.1f TF (%.2f ms)  dominant %s %.0f TF" % (sys.argv[1], d['value'], d['ms_per_step'], r['all_gemm_kernels']['achieved'], r['all_gemm_kernels']['gemm_ms_per_step'], r['kernel'][:40], r['achieved']))
if '-v' in sys.argv[1:]: pass
for k, v in list(r['per_symbol'].items())[:7]:
    print("    %-64s n=%3d %7.3f ms %6.1f TF" % (k[:64], v['launches'], v['ms'], v['tflops']))
PY
done
