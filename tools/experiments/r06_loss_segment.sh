root=$GRAFT_REPO_ROOT
mkdir -p $root/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/ls -o p -- python3 $root/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-mpjpe --no-bf16-legs --no-native-leg --no-collective-path > /tmp/ls.log 2>&1 < /dev/null
python3 $root/tools/segment_between.py /tmp/ls/p_kernel_trace.csv "mesh_att_kernel<2, true>" "mesh_att_bwd1_kernel<2, true>" > $root/gpurun_out/r06_loss_segment.txt 2>&1
python3 $root/tools/segment_between.py /tmp/ls/p_kernel_trace.csv "mesh_gcn_bwd_kernel<0, true>" "bn_maxk_bwd" >> $root/gpurun_out/r06_loss_segment.txt 2>&1
cat $root/gpurun_out/r06_loss_segment.txt
