# batched weight-gradient launch: tiles of one (split, plane) on one XCD (PDF_WG_BATCH_XCD=1, default) vs the 3-D grid (0)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_headline_gpu.py tests/test_ops_gpu.py -x -q -k "heavy or wino or Wino" 2>&1 | tail -3
o=gpurun_out/r04_xcd_ab.txt
: > $o
for v in 1 0 1 0; do
  PDF_WG_BATCH_XCD=$v timeout 600 python bench.py --steps 12 --warmup 5 --no-cpu-baseline --no-mpjpe --no-bf16-legs > /tmp/line.json 2>/tmp/err.txt
  python - "$v" >> $o <<PY
import json, sys
d = json.loads(open('/tmp/line.json').read().strip().splitlines()[-1])
r = d['roofline']
print("PDF_WG_BATCH_XCD=%s : %.1f img/s %.2f ms | %s %.1f TF %.3f ms/step" % (sys.argv[1], d['value'], d['ms_per_step'], r['kernel'], r['achieved'], r['ms_per_step']))
PY
done
cat $o
cd /tmp && export TMPDIR=/tmp
for v in 1 0; do
  export PDF_WG_BATCH_XCD=$v PDF_BENCH_WINOGRAD=1
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/xcd_${v}_$c -o p -- python3 $GRAFT_REPO_ROOT/tools/gemm_bench.py head_3x3 > /tmp/xcd_${v}_${c}.log 2>&1 < /dev/null
  done
  python3 - $v >> $GRAFT_REPO_ROOT/$o <<PY
import csv, sys, collections
v = sys.argv[1]
tot = collections.defaultdict(lambda: [0.0, 0])
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    for r in csv.DictReader(open('/tmp/xcd_%s_%s/p_counter_collection.csv' % (v, c))):
        if 'wgemm_tn_dma' in r['Kernel_Name'] and r['Counter_Name'] == c:
            t = tot[c]; t[0] += float(r['Counter_Value']); t[1] += 1
f, w = tot['FETCH_SIZE'], tot['WRITE_SIZE']
print("PDF_WG_BATCH_XCD=%s head_3x3 (256->256 @64x64, B=32) Winograd weight gradient: FETCH_SIZE x2 KiB->MB %.1f per launch, WRITE_SIZE %.1f MB per launch (%d launches)" % (v, f[0] * 2 * 1024 / max(f[1], 1) / 1e6, w[0] * 1024 / max(w[1], 1) / 1e6, f[1]))
PY
done
tail -2 $GRAFT_REPO_ROOT/$o
