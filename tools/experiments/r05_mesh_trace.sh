#!/bin/bash
# In-step durations of the fused mesh decoder's kernels (rocprofv3 kernel trace of 3 bench steps) next to what runs beside them
root=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/mt -o p -- python3 $root/bench.py --steps 3 --warmup 3 --no-cpu-baseline --no-roofline --no-mpjpe --no-bf16-legs --no-collective-path > /tmp/mt.log 2>&1 < /dev/null
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('/tmp/mt/**/p_kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
mesh = [r for r in rows if 'mesh_' in r['Kernel_Name']]
# last step only: the last 3 x (3 + 9) launches
per = collections.OrderedDict()
for r in mesh[-36:]:
    n = r['Kernel_Name'].split('(')[0].replace('void ', '')
    per.setdefault(n, []).append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for n, v in per.items():
    print("%-48s n=%d  %s us" % (n, len(v), " ".join("%.0f" % x for x in v)))
# what overlapped the level-0 gcn backward kernel (first mesh_gcn_bwd<0,false> of the last step)
tgt = [r for r in mesh[-36:] if 'mesh_gcn_bwd_kernel<0, false>' in r['Kernel_Name']]
if tgt:
    t0, t1 = int(tgt[0]['Start_Timestamp']), int(tgt[0]['End_Timestamp'])
    ov = collections.Counter()
    for r in rows:
        a, b = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        if b > t0 and a < t1 and r is not tgt[0]:
            ov[r['Kernel_Name'].split('(')[0][:60]] += (min(b, t1) - max(a, t0)) / 1e3
    print("overlapping the first level-0 GCN backward launch (%.0f us):" % ((t1 - t0) / 1e3))
    for k, v in ov.most_common(8):
        print("   %-60s %.0f us" % (k, v))
PY
