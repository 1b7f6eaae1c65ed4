# mesh decoder: 8 waves per workgroup (the default build) against 4 (libpdfnet_hip_w4.so, the form of round 5) and 16 (libpdfnet_hip_w16.so, experiment):
# parity, per-kernel durations, the fp32 and bf16 steps
root=$GRAFT_REPO_ROOT
cd $root
mkdir -p gpurun_out
out=$root/gpurun_out/r06_mesh_waves.txt
: > $out
echo "== parity of the default build (8 waves)" >> $out
timeout 900 python -m pytest tests/test_meshdec_gpu.py tests/test_dualgraph_golden_gpu.py tests/test_modules_gpu.py -x -q 2>&1 | tail -3 >> $out
echo "== parity of the 16-wave build" >> $out
PDFNET_HIP_LIB=$root/pdfnet_amd/libpdfnet_hip_w16.so timeout 900 python -m pytest tests/test_meshdec_gpu.py tests/test_dualgraph_golden_gpu.py -x -q 2>&1 | tail -3 >> $out
cd /tmp && export TMPDIR=/tmp
for lib in libpdfnet_hip_w4.so libpdfnet_hip.so libpdfnet_hip_w16.so; do
  export PDFNET_HIP_LIB=$root/pdfnet_amd/$lib
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/mk.$lib -o p -- python3 $root/tools/mesh_bench.py 32 > /tmp/mk.$lib.log 2>&1 < /dev/null
  cp /tmp/mk.$lib/p_kernel_stats.csv $root/gpurun_out/r06_mesh_kernels_$lib.csv
  grep "all levels" /tmp/mk.$lib.log | sed "s/^/$lib (under rocprofv3) /" >> $out
done
python3 - >> $out <<PY
import csv
def load(f):
    return {r['Name']: (int(r['Calls']), float(r['AverageNs']) / 1e3) for r in csv.DictReader(open(f)) if 'mesh_' in r['Name']}
a = load('$root/gpurun_out/r06_mesh_kernels_libpdfnet_hip_w4.so.csv'); b = load('$root/gpurun_out/r06_mesh_kernels_libpdfnet_hip.so.csv'); c = load('$root/gpurun_out/r06_mesh_kernels_libpdfnet_hip_w16.so.csv')
print("# average us per launch: 4 waves -> 8 waves (default) -> 16 waves per workgroup (tools/mesh_bench.py 32 under rocprofv3 --kernel-trace --stats)")
for n in sorted(a, key=lambda n: -a[n][1] * a[n][0]):
    if n in b and n in c:
        print("%-66s calls %4d  %8.1f -> %8.1f -> %8.1f  (%.2fx, %.2fx)" % (n[:66], a[n][0], a[n][1], b[n][1], c[n][1], a[n][1] / b[n][1], a[n][1] / c[n][1]))
PY
cd $root
for r in 1 2; do
for lib in libpdfnet_hip_w4.so libpdfnet_hip.so; do
  export PDFNET_HIP_LIB=$root/pdfnet_amd/$lib
  echo "round $r $lib: fp32 B=32 img/s, ms/step | bf16 B=32 | bf16 B=64" >> $out
  timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-mpjpe --no-roofline --no-bf16-legs --no-native-leg 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" >> $out 2>&1
  for b in 32 64; do timeout 300 python bench.py --dtype bf16 --batch $b --steps 30 --warmup 10 --no-cpu-baseline --no-mpjpe --no-roofline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" >> $out 2>&1; done
done
done
cat $out
