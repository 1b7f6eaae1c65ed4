root=$GRAFT_REPO_ROOT
mkdir -p $root/gpurun_out
cd /tmp && export TMPDIR=/tmp
PDFNET_SIDE_STREAMS=0 timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/lo -o p -- python3 $root/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-mpjpe --no-bf16-legs --no-native-leg --no-collective-path > /tmp/lo.log 2>&1 < /dev/null
head -1 /tmp/lo/p_kernel_trace.csv
python3 $root/tools/low_occupancy.py /tmp/lo/p_kernel_trace.csv > $root/gpurun_out/r06_low_occupancy.txt 2>&1
cat $root/gpurun_out/r06_low_occupancy.txt
