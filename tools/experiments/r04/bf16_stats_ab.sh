# bf16 mode: BatchNorm statistics out of the bf16 GEMM epilogue (PDFNET_BN_EPILOGUE_STATS_BF16=1) vs the BatchNorm's own pass, B=64 and B=32
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
o=gpurun_out/r04_bf16_stats_ab.txt
: > $o
for b in 64 32; do
for v in 0 1 0 1; do
  PDFNET_BN_EPILOGUE_STATS_BF16=$v timeout 600 python bench.py --dtype bf16 --batch $b --steps 10 --warmup 6 --no-cpu-baseline --no-mpjpe > /tmp/line.json 2>/tmp/err.txt
  python - "$v" "$b" >> $o <<PY
import json, sys
d = json.loads(open('/tmp/line.json').read().strip().splitlines()[-1])
print("B=%s PDFNET_BN_EPILOGUE_STATS_BF16=%s : %.1f img/s %.2f ms (%s)" % (sys.argv[2], sys.argv[1], d['value'], d['ms_per_step'], d.get('launch')))
PY
done
done
cat $o
