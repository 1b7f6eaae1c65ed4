cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_bf16_gpu.py tests/test_configs_gpu.py -q 2>&1 | grep -E "passed|failed"
o=gpurun_out/r04_bf16_final_ab.txt
: > $o
for b in 64 32; do
for cfg in "auto auto" "0 0" "auto auto"; do
  set -- $cfg
  PDFNET_BF16_STORAGE=$1 PDFNET_BN_EPILOGUE_STATS_BF16=$2 timeout 600 python bench.py --dtype bf16 --batch $b --steps 10 --warmup 6 --no-cpu-baseline --no-mpjpe > /tmp/line.json 2>/tmp/err.txt
  python - "$1 $2" "$b" >> $o <<PY
import json, sys
d = json.loads(open('/tmp/line.json').read().strip().splitlines()[-1])
print("B=%s storage / epilogue statistics = %s : %.1f img/s %.2f ms" % (sys.argv[2], sys.argv[1], d['value'], d['ms_per_step']))
PY
done
done
cat $o
