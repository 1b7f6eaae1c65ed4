# wider eligibility of the Winograd path: 64-channel layers (PDF_WINOGRAD_MINC=64), ResNet layer 4 (PDF_WINOGRAD_MINPT=4096)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
o=gpurun_out/r04_wino_more_ab.txt
: > $o
for cfg in "128 16384" "64 16384" "128 4096" "64 4096" "128 16384"; do
  set -- $cfg
  PDF_WINOGRAD_MINC=$1 PDF_WINOGRAD_MINPT=$2 timeout 600 python bench.py --steps 12 --warmup 5 --no-cpu-baseline --no-mpjpe --no-bf16-legs > /tmp/line.json 2>/tmp/err.txt
  python - "$1 $2" >> $o <<PY
import json, sys
d = json.loads(open('/tmp/line.json').read().strip().splitlines()[-1])
print("PDF_WINOGRAD_MINC / MINPT = %s : %.1f img/s %.2f ms" % (sys.argv[1], d['value'], d['ms_per_step']))
PY
done
cat $o
