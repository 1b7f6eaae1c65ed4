# F(4x4) for the forward of `feat` too (PDF_WINOGRAD_F4=15) vs the hybrid (11): step time
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
o=gpurun_out/r04_f4_ab.txt
: > $o
for v in 11 15 11 15; do
  PDF_WINOGRAD_F4=$v timeout 600 python bench.py --steps 12 --warmup 5 --no-cpu-baseline --no-mpjpe --no-bf16-legs > /tmp/line.json 2>/tmp/err.txt
  python - "$v" >> $o <<PY
import json, sys
d = json.loads(open('/tmp/line.json').read().strip().splitlines()[-1])
pe = d['roofline']['all_gemm_kernels']['per_entry_point']
print("PDF_WINOGRAD_F4=%s : %.1f img/s %.2f ms | conv2d_fwd %.2f ms" % (sys.argv[1], d['value'], d['ms_per_step'], pe['pdf_conv2d_fwd']['ms']))
PY
done
cat $o
