cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3i
python tools/probe/phase_events.py 32 > gpurun_out/r3i/phase_events.txt 2>&1
cd /tmp && export TMPDIR=/tmp
run="python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-mpjpe --no-bf16-legs"
PDFNET_SIDE_STREAMS=0 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kx -o p -- $run > /tmp/kx.log 2>&1 < /dev/null
cp /tmp/kx/p_kernel_stats.csv $GRAFT_REPO_ROOT/gpurun_out/r3i/kernel_stats_exclusive.csv
tail -16 $GRAFT_REPO_ROOT/gpurun_out/r3i/phase_events.txt
