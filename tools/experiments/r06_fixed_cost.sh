# which kernels do not scale with the batch?  exclusive kernel stats of the fp32 step at B = 32 and B = 64 (side streams off), per symbol: fixed = 2 t(32) - t(64)
root=$GRAFT_REPO_ROOT
mkdir -p $root/gpurun_out
cd /tmp && export TMPDIR=/tmp
for b in 32 64; do
  PDFNET_SIDE_STREAMS=0 timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fc$b -o p -- python3 $root/bench.py --batch $b --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-mpjpe --no-bf16-legs --no-native-leg --no-collective-path > /tmp/fc$b.log 2>&1 < /dev/null
  cp /tmp/fc$b/p_kernel_stats.csv $root/gpurun_out/r06_fixed_cost_B$b.csv
done
python3 - > $root/gpurun_out/r06_fixed_cost.txt <<PY
import csv, re
def load(f):
    d = {}
    for r in csv.DictReader(open(f)):
        n = re.sub(r'\(.*', '', r['Name'])[:70]
        d[n] = d.get(n, 0.0) + float(r['TotalDurationNs']) / 5e6          # 5 traced steps
    return d
a, b = load('$root/gpurun_out/r06_fixed_cost_B32.csv'), load('$root/gpurun_out/r06_fixed_cost_B64.csv')
rows = sorted(((2 * a.get(n, 0) - b.get(n, 0), n) for n in set(a) | set(b)), reverse=True)
print("# exclusive kernel ms per step at B = 32 / B = 64 (fp32, side streams off) and the part that does not scale: fixed = 2 t(32) - t(64)")
print("total: %.2f / %.2f ms, fixed %.2f ms" % (sum(a.values()), sum(b.values()), 2 * sum(a.values()) - sum(b.values())))
for f, n in rows[:45]:
    print("%7.3f fixed | %8.3f %8.3f  %s" % (f, a.get(n, 0), b.get(n, 0), n))
print("...")
for f, n in rows[-8:]:
    print("%7.3f fixed | %8.3f %8.3f  %s" % (f, a.get(n, 0), b.get(n, 0), n))
PY
cat $root/gpurun_out/r06_fixed_cost.txt
