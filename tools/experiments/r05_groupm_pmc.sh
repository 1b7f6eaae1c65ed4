#!/bin/bash
# HBM-side traffic (L2 misses: FETCH_SIZE, doubled per the gfx950 correction) of the igemm_nt symbols per launch in the train step, tile order gm = 1 vs 4
root=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
for gm in 1 4; do
  PDF_IG_GROUPM=$gm PDFNET_SIDE_STREAMS=0 timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/gm$gm -o p -- python3 $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-mpjpe --no-bf16-legs --no-collective-path > /tmp/gm$gm.log 2>&1 < /dev/null
done
python3 - <<PY
import csv, glob, collections, re
res = {}
for gm in (1, 4):
    f = glob.glob('/tmp/gm%d/**/*counter_collection.csv' % gm, recursive=True)[0]
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] != 'FETCH_SIZE' or 'igemm_nt' not in r['Kernel_Name']:
            continue
        k = re.sub(r'^void |\(.*$', '', r['Kernel_Name'])
        agg[k][0] += 1
        agg[k][1] += float(r['Counter_Value']) * 1024 * 2
    res[gm] = agg
print("symbol: launches, read MB per launch at PDF_IG_GROUPM=1 -> 4")
for k in sorted(res[1]):
    a, b = res[1][k], res[4].get(k, [0, 0.0])
    print("%-60s %5d  %8.1f -> %8.1f MB" % (k, a[0] // 3, a[1] / max(a[0], 1) / 1e6, b[1] / max(b[0], 1) / 1e6))
PY
