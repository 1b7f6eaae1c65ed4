# per-kernel durations of the mesh decoder kernels, 4 waves (libpdfnet_hip.so) against 8 waves (libpdfnet_hip_w8.so): rocprofv3 kernel stats of tools/mesh_bench.py
root=$GRAFT_REPO_ROOT
mkdir -p $root/gpurun_out
cd /tmp && export TMPDIR=/tmp
for lib in libpdfnet_hip.so libpdfnet_hip_w8.so; do
  export PDFNET_HIP_LIB=$root/pdfnet_amd/$lib
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/mk.$lib -o p -- python3 $root/tools/mesh_bench.py 32 > /tmp/mk.$lib.log 2>&1 < /dev/null
  cp /tmp/mk.$lib/p_kernel_stats.csv $root/gpurun_out/r06_mesh_kernels_$lib.csv
done
python3 - <<PY
import csv
def load(f):
    return {r['Name']: (int(r['Calls']), float(r['AverageNs']) / 1e3) for r in csv.DictReader(open(f)) if 'mesh_' in r['Name']}
a = load('$root/gpurun_out/r06_mesh_kernels_libpdfnet_hip.so.csv'); b = load('$root/gpurun_out/r06_mesh_kernels_libpdfnet_hip_w8.so.csv')
with open('$root/gpurun_out/r06_mesh_w8_kernels.txt', 'w') as f:
    f.write("# average us per launch: 4 waves -> 8 waves per workgroup (tools/mesh_bench.py 32 under rocprofv3 --kernel-trace --stats)\n")
    for n in sorted(a, key=lambda n: -a[n][1] * a[n][0]):
        if n in b:
            f.write("%-70s calls %4d  %8.1f -> %8.1f  (%.2fx)\n" % (n[:70], a[n][0], a[n][1], b[n][1], a[n][1] / b[n][1]))
print(open('$root/gpurun_out/r06_mesh_w8_kernels.txt').read())
PY
