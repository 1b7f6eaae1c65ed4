root=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
for v in 0 1; do
PDFNET_SIDE_STREAMS=0 PDFNET_BF16_SHADOWS=$v timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sh$v -o p -- python3 $root/bench.py --dtype bf16 --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-mpjpe > /tmp/sh$v.log 2>&1 < /dev/null
cp /tmp/sh$v/p_kernel_stats.csv $root/gpurun_out/bf16_shadow${v}_stats.csv
done
