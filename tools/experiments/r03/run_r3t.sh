cd /tmp && export TMPDIR=/tmp
root=$GRAFT_REPO_ROOT
timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -o p -- python3 $root/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-mpjpe --no-bf16-legs > /tmp/kt.log 2>&1 < /dev/null
mkdir -p $root/gpurun_out/r3t
f=$(ls /tmp/kt/*kernel_trace.csv /tmp/kt/*/*kernel_trace.csv 2>/dev/null | head -1)
python3 - "$f" $root/gpurun_out/r3t/trace_small.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
keep = ['Kernel_Name', 'Start_Timestamp', 'End_Timestamp', 'Stream_Id', 'Queue_Id', 'Grid_Size', 'Workgroup_Size', 'LDS_Block_Size', 'VGPR_Count']
keep = [k for k in keep if k in rows[0]]
w = csv.writer(open(sys.argv[2], 'w'))
w.writerow(keep)
for r in rows:
    w.writerow([r[k][:110] if k == 'Kernel_Name' else r[k] for k in keep])
PY
gzip -f $root/gpurun_out/r3t/trace_small.csv
ls -la $root/gpurun_out/r3t/
tail -2 /tmp/kt.log | cut -c1-300
