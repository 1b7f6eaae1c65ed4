cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
out=gpurun_out/r06_fork_dense_loss2.txt
: > $out
for r in 4 5 6 7 8; do for v in 0 1; do
  echo "round $r PDFNET_FORK_DENSE_LOSS=$v: $(PDFNET_FORK_DENSE_LOSS=$v timeout 300 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-mpjpe --no-roofline --no-bf16-legs --no-native-leg 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['median_step_ms'])")" >> $out
done; done
cat $out
