# exclusive per-kernel times of the step with and without x3 (side streams off), 5 traced steps each
root=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
run="python3 $root/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-mpjpe --no-bf16-legs --no-collective-path"
for x in 0 1; do
  PDF_X3=$x PDFNET_SIDE_STREAMS=0 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/x3k$x -o p -- $run > /tmp/x3k$x.log 2>&1 < /dev/null
  cp /tmp/x3k$x/p_kernel_stats.csv $root/gpurun_out/r06_x3_${x}_kernel_stats_exclusive.csv
done
