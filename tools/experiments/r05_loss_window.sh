#!/bin/bash
# What runs between the mesh decoder's last forward level and its first backward level (decoder tail, loss forward, loss backward, tail backward)
root=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/mt -o p -- python3 $root/bench.py --steps 3 --warmup 3 --no-cpu-baseline --no-roofline --no-mpjpe --no-bf16-legs --no-collective-path > /tmp/mt.log 2>&1 < /dev/null
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('/tmp/mt/**/p_kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
print(list(rows[0].keys()))
idx_f = max(i for i, r in enumerate(rows) if 'mesh_att_kernel<2, true>' in r['Kernel_Name'])
idx_b = max(i for i, r in enumerate(rows) if 'mesh_att_bwd1_kernel<2, true>' in r['Kernel_Name'])
q = rows[idx_f].get('Queue_Id')
t0 = int(rows[idx_f]['End_Timestamp'])
print("window %.1f us, %d kernels (all queues), main queue %s" % ((int(rows[idx_b]['Start_Timestamp']) - t0) / 1e3, idx_b - idx_f - 1, q))
prev = t0
agg = collections.OrderedDict()
for r in rows[idx_f + 1:idx_b]:
    if r.get('Queue_Id') != q:
        continue
    a, b = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    n = r['Kernel_Name'].split('(')[0].replace('void ', '')[:70]
    print("%8.1f  gap %6.1f  dur %6.1f  %s" % ((a - t0) / 1e3, (a - prev) / 1e3, (b - a) / 1e3, n))
    prev = b
PY
