cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3u
timeout 600 python bench.py --dtype bf16 --batch 64 --steps 10 --warmup 6 --no-cpu-baseline --no-mpjpe --gemm-shapes gpurun_out/r3u/shapes_bf16_B64.txt > gpurun_out/r3u/b_bf16_B64.json 2>gpurun_out/r3u/err.txt
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r3u/b_bf16_B64.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'])
r = d['roofline']
for k, v in r['per_symbol'].items() if isinstance(r['per_symbol'], dict) else [(x.get('kernel'), x) for x in r['per_symbol']]:
    print(k, v)
print(r['all_gemm_kernels']['achieved'], r['all_gemm_kernels']['gemm_ms_per_step'])
for k, v in r['all_gemm_kernels']['per_entry_point'].items(): print(k, v)
PY
head -60 gpurun_out/r3u/shapes_bf16_B64.txt
