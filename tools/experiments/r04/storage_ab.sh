# bf16 B=64: storage of the conv -> BatchNorm tensors off / on (the LDS-DMA kernel now writes the bf16 output itself)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_bf16_gpu.py tests/test_configs_gpu.py tests/test_model_gpu.py -q -x 2>&1 | grep -E "passed|failed"
o=gpurun_out/r04_storage_ab.txt
: > $o
for v in 0 1 0 1; do
  PDFNET_BF16_STORAGE=$v timeout 600 python bench.py --dtype bf16 --batch 64 --steps 10 --warmup 6 --no-cpu-baseline --no-mpjpe > /tmp/line.json 2>/tmp/err.txt
  python - "$v" >> $o <<PY
import json, sys
d = json.loads(open('/tmp/line.json').read().strip().splitlines()[-1])
r = d['roofline']
print("PDFNET_BF16_STORAGE=%s : %.1f img/s %.2f ms (%s) | %s %d launches %.2f ms %.0f TF | gemm %.1f ms" % (sys.argv[1], d['value'], d['ms_per_step'], d.get('launch'), r['kernel'], r['launches_per_step'], r['ms_per_step'], r['achieved'], r['all_gemm_kernels']['gemm_ms_per_step']))
PY
done
cat $o
