# F(4x4) everywhere (15) with / without ResNet layer 3 under the Winograd path (PDF_WINOGRAD_MINPT=16384): parity margins, step time
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
o=gpurun_out/r04_f4_l3_ab.txt
: > $o
export PDF_WINOGRAD_F4=15
for m in 65536 16384; do
  echo "== PDF_WINOGRAD_F4=15 PDF_WINOGRAD_MINPT=$m" >> $o
  PDF_WINOGRAD_MINPT=$m python -m pytest tests/test_headline_gpu.py -x -q -s -k "train_step_at_the_headline or eval_forward_at" 2>&1 | grep -E "worst five|passed|failed" | cut -c1-700 >> $o
done
for m in 65536 16384 65536 16384; do
  PDF_WINOGRAD_MINPT=$m timeout 600 python bench.py --steps 12 --warmup 5 --no-cpu-baseline --no-mpjpe --no-bf16-legs > /tmp/line.json 2>/tmp/err.txt
  python - "$m" >> $o <<PY
import json, sys
d = json.loads(open('/tmp/line.json').read().strip().splitlines()[-1])
print("PDF_WINOGRAD_F4=15 PDF_WINOGRAD_MINPT=%s : %.1f img/s %.2f ms" % (sys.argv[1], d['value'], d['ms_per_step']))
PY
done
cat $o
