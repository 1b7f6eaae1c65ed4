cd /tmp && export TMPDIR=/tmp
root=$GRAFT_REPO_ROOT
PDFNET_BF16_STORAGE=1 PDFNET_SIDE_STREAMS=0 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kx16 -o p -- python3 $root/bench.py --dtype bf16 --batch 64 --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-mpjpe > /tmp/kx16.log 2>&1 < /dev/null
cp /tmp/kx16/p_kernel_stats.csv $root/gpurun_out/r03_kernel_stats_exclusive_bf16_B64_storage.csv
tail -1 /tmp/kx16.log | cut -c1-200
