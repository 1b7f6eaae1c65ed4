# mesh loss kernels on 1,024 threads per (hand, sample) block (were 256): parity, kernel durations in the step; the transposed convolutions' weight-gradient tile rule: parity
root=$GRAFT_REPO_ROOT
cd $root
mkdir -p gpurun_out
out=$root/gpurun_out/r06_mesh_loss_threads.txt
: > $out
timeout 900 python -m pytest tests/test_loss_gpu.py tests/test_trainer_gpu.py -x -q 2>&1 | tail -2 >> $out
timeout 900 python -m pytest tests/test_headline_gpu.py -x -q 2>&1 | tail -2 >> $out
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ml -o p -- python3 $root/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-mpjpe --no-bf16-legs --no-native-leg --no-collective-path > /tmp/ml.log 2>&1 < /dev/null
grep "mesh_loss\|x3gemm_tn" /tmp/ml/p_kernel_stats.csv | cut -d, -f1-4 >> $out
cd $root
for r in 1 2 3; do
  timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-mpjpe --no-roofline --no-bf16-legs --no-native-leg 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" >> $out 2>&1
done
cat $out
