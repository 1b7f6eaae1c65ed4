"""Kernel census of time windows inside the last train step of a rocprofv3 --kernel-trace CSV (plain or .gz).
Usage: python tools/window_kernels.py <trace.csv[.gz]> <from_ms> <to_ms> [top]
Times are ms since the first kernel of the step.  Prints per-stream busy time, launches, idle time of the union, and the
kernels with the most time in the window."""
import csv, sys, gzip, collections, re

path, a, b = sys.argv[1], float(sys.argv[2]), float(sys.argv[3])
top = int(sys.argv[4]) if len(sys.argv) > 4 else 25
op = gzip.open if path.endswith('.gz') else open
rows = list(csv.DictReader(op(path, 'rt')))
ks = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Stream_Id', r.get('Queue_Id', '0')), r.get('Grid_Size', '?'), r.get('Workgroup_Size', '?')) for r in rows)
adams = [i for i, k in enumerate(ks) if k[2].startswith('adam_kernel')]
if len(adams) >= 2:
    gaps_a = [ks[adams[i + 1]][0] - ks[adams[i]][0] for i in range(len(adams) - 1)]
    adams = [adams[i] for i in range(len(adams) - 1) if gaps_a[i] > 0.5 * max(gaps_a)] + [adams[-1]]
step = ks[adams[-2] + 1:adams[-1] + 1]
t0 = step[0][0]
print("step: %d kernels, %.2f ms" % (len(step), (max(k[1] for k in step) - t0) / 1e6))
lo, hi = t0 + a * 1e6, t0 + b * 1e6
win = [k for k in step if k[0] >= lo and k[0] < hi]
per = collections.defaultdict(lambda: [0, 0])
for s, e, n, q, g, w in win:
    per[q][0] += 1; per[q][1] += e - s
for q, (c, d) in sorted(per.items(), key=lambda x: -x[1][1]):
    print("stream %-4s %5d launches  busy %7.3f ms" % (q, c, d / 1e6))
busy = 0; cs, ce = win[0][0], win[0][1]
for s, e, *_ in win[1:]:
    if s > ce:
        busy += ce - cs; cs, ce = s, e
    else:
        ce = max(ce, e)
busy += ce - cs
print("window %.2f-%.2f ms: %d launches, union busy %.3f ms, idle %.3f ms" % (a, b, len(win), busy / 1e6, (b - a) - busy / 1e6))
agg = collections.defaultdict(lambda: [0, 0, set()])
for s, e, n, q, g, w in win:
    n = re.sub(r"^void |\(.*$", "", n)[:70]
    agg[n][0] += 1; agg[n][1] += e - s
    try:
        agg[n][2].add(int(g) // max(int(w), 1))
    except ValueError:
        pass
for n, (c, d, gs) in sorted(agg.items(), key=lambda x: -x[1][1])[:top]:
    gl = sorted(gs)
    print("%5d x %8.1f us avg  %7.3f ms  blocks %s..%s  %s" % (c, d / c / 1e3, d / 1e6, gl[0] if gl else '?', gl[-1] if gl else '?', n))
