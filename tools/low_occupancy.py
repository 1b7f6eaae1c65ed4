"""Kernels of one traced step that cannot fill the chip: per symbol, the time spent in launches with fewer workgroups than CUs (256) or fewer
waves than 2 per SIMD (2,048), from a rocprofv3 --kernel-trace CSV.   python tools/low_occupancy.py <kernel_trace.csv>"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted(rows, key=lambda r: int(r['Start_Timestamp']))
adams = [i for i, r in enumerate(ks) if r['Kernel_Name'].startswith('adam_kernel')]
if len(adams) >= 2:
    ks = ks[adams[-2] + 1:adams[-1] + 1]


def n(r, k):
    return int(r.get(k) or 1)


tot = collections.defaultdict(lambda: [0, 0.0, 0, 0.0, 0, 0])
for r in ks:
    wgs = 1
    thr = 1
    for a in 'XYZ':
        g, w = n(r, 'Grid_Size_' + a), n(r, 'Workgroup_Size_' + a)
        wgs *= max(1, g // max(1, w))
        thr *= w
    waves = wgs * ((thr + 63) // 64)
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    t = tot[r['Kernel_Name'][:84]]
    t[0] += 1
    t[1] += d
    if wgs < 256 or waves < 2048:
        t[2] += 1
        t[3] += d
        t[4] = max(t[4], wgs)
        t[5] = max(t[5], waves)
step = sum(t[1] for t in tot.values())
low = sum(t[3] for t in tot.values())
print("kernel time in the step %.2f ms; in launches with < 256 workgroups or < 2,048 waves: %.2f ms" % (step / 1e3, low / 1e3))
print("%-84s %6s %9s | %6s %9s  max WGs  max waves" % ("kernel", "calls", "us", "low", "us"))
for k, t in sorted(tot.items(), key=lambda kv: -kv[1][3])[:40]:
    if t[3] > 0:
        print("%-84s %6d %9.1f | %6d %9.1f  %7d  %9d" % (k, t[0], t[1], t[2], t[3], t[4], t[5]))
