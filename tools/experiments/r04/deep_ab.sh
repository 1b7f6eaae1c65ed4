# (historical: the PDF_IG_DEEP variants this script drives were removed after the measurement -- profiles/r04_igemm_dma_ab.txt)
# deep-ring LDS-DMA kernel on the latency-bound small products (mesh decoder, centre windows): step time and the pair entry points
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
o=gpurun_out/r04_deep_ab.txt
: > $o
for cfg in "0 512" "1 512" "2 512" "3 512" "1 1024" "2 1024" "0 512"; do
  set -- $cfg
  PDF_IG_DEEP=$1 PDF_IG_DEEP_MAXT=$2 timeout 600 python bench.py --steps 12 --warmup 5 --no-cpu-baseline --no-mpjpe --no-bf16-legs > /tmp/line.json 2>/tmp/err.txt
  python - "$1 $2" >> $o <<PY
import json, sys
d = json.loads(open('/tmp/line.json').read().strip().splitlines()[-1])
pe = d['roofline']['all_gemm_kernels']['per_entry_point']
print("PDF_IG_DEEP / MAXT = %s : %.1f img/s %.2f ms | " % (sys.argv[1], d['value'], d['ms_per_step']) + "  ".join("%s %.2f" % (k.replace('pdf_', ''), pe[k]['ms']) for k in sorted(pe) if 'pair' in k or 'linear' in k))
PY
done
cat $o
bash tools/experiments/r04/bn_ab3.sh > /dev/null 2>&1
grep -v amdgpu gpurun_out/r04_bn_ab3.txt | grep -E "==|all:|C=64 R=1048576"
