cd /tmp && export TMPDIR=/tmp
root=$GRAFT_REPO_ROOT
for st in 0 1; do
PDFNET_BF16_STORAGE=$st PDFNET_SIDE_STREAMS=0 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kx32_$st -o p -- python3 $root/bench.py --dtype bf16 --batch 32 --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-mpjpe > /tmp/kx32_$st.log 2>&1 < /dev/null
cp /tmp/kx32_$st/p_kernel_stats.csv $root/gpurun_out/kx32_$st.csv
done
