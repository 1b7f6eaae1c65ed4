cd /tmp && export TMPDIR=/tmp
PDFNET_SIDE_STREAMS=0 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kx -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-mpjpe --no-bf16-legs > /tmp/kx.log 2>&1 < /dev/null
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open('/tmp/kx/p_kernel_stats.csv')))
steps=5
tot=sum(float(r['TotalDurationNs']) for r in rows)
print("total ms per step %.2f"%(tot/1e6/steps))
for r in rows[:24]:
    print("%-80s calls %6.1f %8.3f ms"%(r['Name'][:80], int(r['Calls'])/steps, float(r['TotalDurationNs'])/1e6/steps))
PY
