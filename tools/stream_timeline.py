"""Per-stream view of one train step from a rocprofv3 --kernel-trace CSV: busy time per HIP stream, how much of the step
has 1 / 2 / 3+ kernels resident, and the kernel classes that fill the busiest stream.
Usage: python tools/stream_timeline.py <kernel_trace.csv>"""
import csv, sys, collections, re

rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Stream_Id', r.get('Queue_Id', '0'))) for r in rows)
adams = [i for i, k in enumerate(ks) if k[2].startswith('adam_kernel')]
step = ks[adams[-2] + 1:adams[-1] + 1]
t0, t1 = step[0][0], max(k[1] for k in step)
print("kernels %d, wall %.2f ms, sum of durations %.2f ms" % (len(step), (t1 - t0) / 1e6, sum(e - s for s, e, _, _ in step) / 1e6))


def cls(n):
    for pat, c in (('wgemm', 'wgemm'), ('igemm', 'igemm'), ('reduce_slabs', 'reduce_slabs'), ('bn_', 'batchnorm'), ('affine_apply', 'batchnorm'),
                   ('layernorm', 'layernorm'), ('attn_', 'attention'), ('maxk', 'maxk'), ('knn', 'knn/group'), ('group_bwd', 'knn/group'),
                   ('colsum', 'colsum'), ('at::native', 'aten'), ('cheby', 'cheby'), ('adam', 'adam'), ('small_k', 'igemm')):
        if pat in n:
            return c
    return 'other hip'


per = collections.defaultdict(lambda: [0, 0, collections.Counter()])
for s, e, n, q in step:
    per[q][0] += e - s
    per[q][1] += 1
    per[q][2][cls(n)] += e - s
for q, (b, c, cc) in sorted(per.items(), key=lambda kv: -kv[1][0]):
    print("stream %-6s busy %7.2f ms  %5d kernels   %s" % (q, b / 1e6, c, ', '.join("%s %.1f" % (k, v / 1e6) for k, v in cc.most_common(7))))
# concurrency histogram
ev = []
for s, e, n, q in step:
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
lvl, last, hist = 0, t0, collections.Counter()
for t, d in ev:
    hist[min(lvl, 3)] += t - last
    last = t
    lvl += d
print("resident kernels: " + ', '.join("%s: %.2f ms" % (('0', '1', '2', '3+')[k], v / 1e6) for k, v in sorted(hist.items())))
# time with NO gemm-class kernel resident
ev = []
for s, e, n, q in step:
    if cls(n) in ('wgemm', 'igemm'):
        ev.append((s, 1)); ev.append((e, -1))
ev.sort()
lvl, last, nog = 0, t0, 0
for t, d in ev:
    if lvl == 0:
        nog += t - last
    last = t
    lvl += d
nog += t1 - last
print("time with no GEMM kernel resident: %.2f ms of %.2f" % (nog / 1e6, (t1 - t0) / 1e6))
