"""Per-stream view of one train step from a rocprofv3 --kernel-trace CSV: busy time per HIP stream, how much of the step
has 1 / 2 / 3+ kernels resident, and the kernel classes that fill the busiest stream.
Usage: python tools/stream_timeline.py <kernel_trace.csv>"""
import csv, sys, collections, re

rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Stream_Id', r.get('Queue_Id', '0'))) for r in rows)
adams = [i for i, k in enumerate(ks) if k[2].startswith('adam_kernel')]
# an optimizer step may launch Adam more than once (early slice beside the backward): a step ends at an Adam launch that is
# followed by a long Adam-free stretch (or by nothing)
if len(adams) >= 2:
    gaps_a = [ks[adams[i + 1]][0] - ks[adams[i]][0] for i in range(len(adams) - 1)]
    ends = [adams[i] for i in range(len(adams) - 1) if gaps_a[i] > 0.5 * max(gaps_a)] + [adams[-1]]
    adams = ends
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

# ---- the busiest stream's idle gaps: where the dependent chain waits (for a forked branch, the weight-gradient stream, the host)
main = max(per.items(), key=lambda kv: kv[1][0])[0]
mk = [k for k in step if k[3] == main]
others = [k for k in step if k[3] != main]
gaps = []
for a, b in zip(mk, mk[1:]):
    if b[0] - a[1] >= 20000:
        busy = sum(max(0, min(o[1], b[0]) - max(o[0], a[1])) for o in others)
        gaps.append((b[0] - a[1], (a[1] - t0) / 1e6, a[2][:40], b[2][:40], busy / (b[0] - a[1])))
print("stream %s idle gaps >= 20 us: %d, total %.2f ms (all gaps %.2f ms)" % (main, len(gaps), sum(g[0] for g in gaps) / 1e6,
      sum(b[0] - a[1] for a, b in zip(mk, mk[1:]) if b[0] > a[1]) / 1e6))
for g in sorted(gaps, reverse=True)[:25]:
    print("  at %6.2f ms  gap %7.1f us  other streams busy x%.2f   after %-40s before %s" % (g[1], g[0] / 1e3, g[4], g[2], g[3]))
# 4 ms buckets: busy fraction of the main stream and of everything else
nb = int((t1 - t0) / 4e6) + 1
bm, bo = [0] * nb, [0] * nb
for s, e, n, q in step:
    tgt = bm if q == main else bo
    b = int((s - t0) / 4e6)
    while s < e:
        end = min(e, t0 + (b + 1) * 4000000)
        tgt[b] += end - s
        s = end
        b += 1
print("4 ms buckets, main stream busy: " + ' '.join("%.2f" % (v / 4e6) for v in bm))
print("4 ms buckets, other streams  : " + ' '.join("%.2f" % (v / 4e6) for v in bo))
# small side streams kernel by kernel (forked branches: what the main stream's joins wait for)
for q, (b, c, cc) in sorted(per.items(), key=lambda kv: -kv[1][0]):
    if q == main or c > 70:
        continue
    print("stream %s:" % q)
    for s, e, n, qq in step:
        if qq == q and e - s >= 30000:
            print("   at %6.2f ms  %8.1f us  %s" % ((s - t0) / 1e6, (e - s) / 1e3, n[:100]))
# aten kernels on the main stream (glue the library could absorb), by kernel name
ac = collections.defaultdict(lambda: [0, 0])
for s, e, n, q in mk:
    if cls(n) == 'aten' or 'rocclr' in n:
        key = re.sub(r'\s+', ' ', n)[:150]
        ac[key][0] += 1; ac[key][1] += e - s
print("aten / runtime-copy kernels on stream %s: %d launches, %.2f ms" % (main, sum(v[0] for v in ac.values()), sum(v[1] for v in ac.values()) / 1e6))
for k, v in sorted(ac.items(), key=lambda kv: -kv[1][1])[:14]:
    print("  %4d  %7.3f ms  %s" % (v[0], v[1] / 1e6, k))
# the largest of them one by one
big = sorted(((e - s, (s - t0) / 1e6, n) for s, e, n, q in mk if (cls(n) == 'aten' or 'rocclr' in n) and e - s >= 40000), reverse=True)
for d, at, n in big[:40]:
    print("     at %6.2f ms  %7.1f us  %s" % (at, d / 1e3, re.sub(r'\s+', ' ', n)[:110]))
# main-stream kernels that are neither GEMM nor BatchNorm nor aten, by name (the latency-bound part of the dependent chain)
oc = collections.defaultdict(lambda: [0, 0])
for s, e, n, q in mk:
    if cls(n) in ('other hip', 'attention', 'layernorm', 'cheby', 'maxk', 'knn/group', 'colsum'):
        key = re.sub(r'\(.*', '', n)[:60]
        oc[key][0] += 1; oc[key][1] += e - s
print("other library kernels on stream %s: %d launches, %.2f ms" % (main, sum(v[0] for v in oc.values()), sum(v[1] for v in oc.values()) / 1e6))
for k, v in sorted(oc.items(), key=lambda kv: -kv[1][1])[:40]:
    print("  %4d  %7.3f ms  avg %6.1f us  %s" % (v[0], v[1] / 1e6, v[1] / v[0] / 1e3, k))
