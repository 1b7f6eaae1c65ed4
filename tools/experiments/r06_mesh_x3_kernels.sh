# per-kernel durations of the mesh decoder, native fp32 MFMA build (PDF_X3_MESH=0) against the x3 build (default): tools/mesh_bench.py under rocprofv3
root=$GRAFT_REPO_ROOT
mkdir -p $root/gpurun_out
cd /tmp && export TMPDIR=/tmp
for m in 0 1; do
  export PDF_X3_MESH=$m
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/mx$m -o p -- python3 $root/tools/mesh_bench.py 32 > /tmp/mx$m.log 2>&1 < /dev/null
  cp /tmp/mx$m/p_kernel_stats.csv $root/gpurun_out/r06_mesh_x3_kernels_$m.csv
done
unset PDF_X3_MESH
python3 - > $root/gpurun_out/r06_mesh_x3_kernels.txt <<PY
import csv, re
def load(f):
    d = {}
    for r in csv.DictReader(open(f)):
        if 'mesh_' in r['Name']:
            n = re.sub(r'(md_x3_build::|void )', '', r['Name'])
            n = re.sub(r'\(.*', '', n)
            d[n] = (int(r['Calls']), float(r['AverageNs']) / 1e3)
    return d
a, b = load('$root/gpurun_out/r06_mesh_x3_kernels_0.csv'), load('$root/gpurun_out/r06_mesh_x3_kernels_1.csv')
print("# average us per launch: native fp32 MFMA build -> x3 build (tools/mesh_bench.py 32 under rocprofv3 --kernel-trace --stats)")
ta = tb = 0.0
for n in sorted(a, key=lambda n: -a[n][1] * a[n][0]):
    if n in b:
        print("%-50s calls %4d  %8.1f -> %8.1f  (%.2fx)" % (n[:50], a[n][0], a[n][1], b[n][1], a[n][1] / b[n][1]))
PY
cat $root/gpurun_out/r06_mesh_x3_kernels.txt
cd $root
timeout 1200 python -m pytest tests/test_headline_gpu.py tests/test_model_gpu.py tests/test_full_gradient_gpu.py -x -q 2>&1 | grep "passed\|failed" | tail -3
