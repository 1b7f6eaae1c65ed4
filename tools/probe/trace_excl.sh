# exclusive (single-stream) kernel trace of one step: the sequence of kernels with durations -> gpurun_out/<tag>_seq.txt
tag=${1:-seq}
root=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
PDFNET_SIDE_STREAMS=0 timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/$tag.kt -o p -- python3 $root/bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-roofline --no-mpjpe > /tmp/$tag.log 2>&1 < /dev/null
python3 - <<PY > $root/gpurun_out/${tag}_seq.txt
import csv, re
rows = list(csv.DictReader(open('/tmp/$tag.kt/p_kernel_trace.csv')))
ks = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows)
ad = [i for i, k in enumerate(ks) if k[2].startswith('adam_kernel')]
step = ks[ad[-2] + 1:ad[-1] + 1]
t0 = step[0][0]
prev = t0
for s, e, n in step:
    n = re.sub(r'\(.*', '', n)
    n = re.sub(r'void |at::native::|\(anonymous namespace\)::', '', n)[:70]
    print("%9.3f  gap %6.1f  dur %7.1f  %s" % ((s - t0) / 1e6, (s - prev) / 1e3, (e - s) / 1e3, n))
    prev = e
PY
