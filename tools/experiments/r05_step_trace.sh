#!/bin/bash
# Kernel sequence of ONE bench step (all queues): start (us since the step's first kernel), duration, queue, name -> gpurun_out/r05_step_trace.txt
root=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/mt -o p -- python3 $root/bench.py --steps 3 --warmup 3 --no-cpu-baseline --no-roofline --no-mpjpe --no-bf16-legs --no-collective-path > /tmp/mt.log 2>&1 < /dev/null
python3 - <<PY
import csv, glob
f = glob.glob('/tmp/mt/**/p_kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
adam = [i for i, r in enumerate(rows) if 'adam_kernel' in r['Kernel_Name']]
a, b = adam[-2] + 1, adam[-1] + 1
t0 = int(rows[a]['Start_Timestamp'])
qs = {}
with open('$root/gpurun_out/r05_step_trace.txt', 'w') as o:
    for r in rows[a:b]:
        q = qs.setdefault(r['Queue_Id'], len(qs))
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        o.write("%9.1f %7.1f q%d %s\n" % ((s - t0) / 1e3, (e - s) / 1e3, q, r['Kernel_Name'].split('(')[0].replace('void ', '')[:90]))
print(b - a, "kernels in the step;", (int(rows[b - 1]['End_Timestamp']) - t0) / 1e3, "us")
PY
