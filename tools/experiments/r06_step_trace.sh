# exclusive per-kernel times of the shipped fp32 step (side streams off), 5 traced steps
root=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
run="python3 $root/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-mpjpe --no-bf16-legs --no-native-leg --no-collective-path"
PDFNET_SIDE_STREAMS=0 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/k6 -o p -- $run > /tmp/k6.log 2>&1 < /dev/null
cp /tmp/k6/p_kernel_stats.csv $root/gpurun_out/r06_mid_kernel_stats_exclusive.csv
