"""Kernels of one traced step between the last launch of symbol A and the first launch of symbol B after it, in start order, with stream, duration
and the gap to the previous kernel's end on ANY stream.   python tools/segment_between.py <kernel_trace.csv> <A substring> <B substring>"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted(rows, key=lambda r: int(r['Start_Timestamp']))
adams = [i for i, r in enumerate(ks) if r['Kernel_Name'].startswith('adam_kernel')]
if len(adams) >= 2:
    ks = ks[adams[-2] + 1:adams[-1] + 1]
A, B = sys.argv[2], sys.argv[3]
ia = max(i for i, r in enumerate(ks) if A in r['Kernel_Name'])
ib = min(i for i, r in enumerate(ks) if i > ia and B in r['Kernel_Name'])
t0 = int(ks[ia]['End_Timestamp'])
print("from the end of %s to the start of %s: %.1f us, %d kernels" % (A, B, (int(ks[ib]['Start_Timestamp']) - t0) / 1e3, ib - ia - 1))
end = t0
busy = 0.0
for r in ks[ia + 1:ib + 1]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    g = r.get('Grid_Size_X') or r.get('Grid_Size') or '?'
    w = r.get('Workgroup_Size_X') or r.get('Workgroup_Size') or '?'
    print("%9.1f  q%-3s %8.1f us  gap %7.1f  grid %8s wg %5s  %s" % ((s - t0) / 1e3, r.get('Queue_Id', '?')[-3:], (e - s) / 1e3, (s - end) / 1e3, g, w, r['Kernel_Name'][:100]))
    end = max(end, e)
