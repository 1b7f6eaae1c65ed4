"""GPU busy/idle analysis of a rocprofv3 --kernel-trace CSV (one process, eager steps).
Usage: python tools/gap_analysis.py <kernel_trace.csv> [steps_in_trace]
Prints: wall span of the steady steps, union-of-kernels busy time, idle time, idle grouped by the kernel that precedes the
gap, and busy time while only small kernels (< 15 us) are resident."""
import csv, sys, collections

path = sys.argv[1]
rows = list(csv.DictReader(open(path)))
ks = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Queue_Id', '0')) for r in rows))
# steady part: from the second-to-last adam_kernel to the last one = one full step
adams = [i for i, k in enumerate(ks) if k[2].startswith('adam_kernel')]
# an optimizer step may launch Adam more than once (early slice beside the backward): a step ends at an Adam launch that is
# followed by a long Adam-free stretch (or by nothing)
if len(adams) >= 2:
    gaps_a = [ks[adams[i + 1]][0] - ks[adams[i]][0] for i in range(len(adams) - 1)]
    ends = [adams[i] for i in range(len(adams) - 1) if gaps_a[i] > 0.5 * max(gaps_a)] + [adams[-1]]
    adams = ends
if len(adams) < 2:
    print("need >= 2 steps in the trace"); sys.exit(1)
lo, hi = adams[-2] + 1, adams[-1] + 1
step = ks[lo:hi]
t0, t1 = step[0][0], max(k[1] for k in step)
print("kernels in step: %d   wall %.3f ms" % (len(step), (t1 - t0) / 1e6))
# union busy
busy = 0; cur_s, cur_e = step[0][0], step[0][1]
gaps = collections.Counter(); gapn = collections.Counter()
last_name = step[0][2]
for s, e, n, q in step[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        g = s - cur_e
        gaps[last_name[:60]] += g; gapn[last_name[:60]] += 1
        cur_s, cur_e = s, e; last_name = n
    else:
        if e > cur_e:
            cur_e = e; last_name = n
busy += cur_e - cur_s
print("busy (union) %.3f ms   idle %.3f ms (%.1f%%)" % (busy / 1e6, (t1 - t0 - busy) / 1e6, 100.0 * (t1 - t0 - busy) / (t1 - t0)))
hist = collections.Counter()
for k, v in gaps.items():
    pass
allg = []
cur_e = step[0][1]
for s, e, n, q in step[1:]:
    if s > cur_e:
        allg.append(s - cur_e)
    cur_e = max(cur_e, e)
for lim in (2000, 5000, 10000, 20000, 50000, 1e12):
    sel = [g for g in allg if g < lim]
    print("  gaps < %8.0f ns: %5d  sum %.3f ms" % (lim, len(sel), sum(sel) / 1e6))
print("idle by preceding kernel:")
for k, v in gaps.most_common(25):
    print("  %8.3f ms  %5d gaps  avg %6.1f us  %s" % (v / 1e6, gapn[k], v / gapn[k] / 1e3, k))
# time split: sum of durations by kernel (for reference) within the step
dur = collections.Counter()
for s, e, n, q in step:
    dur[n[:60]] += e - s
print("sum of kernel durations in step %.3f ms (overlap counted twice)" % (sum(dur.values()) / 1e6))
# 10 ms timeline buckets: busy fraction
B = 5_000_000
nb = int((t1 - t0) / B) + 1
bb = [0] * nb
cur_s, cur_e = None, None
iv = []
cs, ce = step[0][0], step[0][1]
for s, e, n, q in step[1:]:
    if s > ce:
        iv.append((cs, ce)); cs, ce = s, e
    else:
        ce = max(ce, e)
iv.append((cs, ce))
for s, e in iv:
    b = int((s - t0) / B)
    while s < e:
        lim = t0 + (b + 1) * B
        seg = min(e, lim) - s
        bb[b] += seg; s += seg; b += 1
print("busy fraction per 5 ms bucket:", ' '.join("%.2f" % (x / B) for x in bb))
