# the dense-map loss terms (and their backward) on a forked stream (default) against in line on the main stream (PDFNET_FORK_DENSE_LOSS=0): parity, phases, step
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
out=gpurun_out/r06_fork_dense_loss.txt
: > $out
timeout 900 python -m pytest tests/test_loss_gpu.py tests/test_trainer_gpu.py tests/test_configs_gpu.py -x -q 2>&1 | grep "passed\|failed" | tail -2 >> $out
for v in 0 1; do
  echo "== PDFNET_FORK_DENSE_LOSS=$v" >> $out
  PDFNET_FORK_DENSE_LOSS=$v timeout 300 python tools/probe/phase_events.py 32 2>&1 | grep "fwd: loss\|loss done\|decoder tail done" | tail -3 >> $out
done
for r in 1 2 3; do for v in 0 1; do
  echo "round $r PDFNET_FORK_DENSE_LOSS=$v: $(PDFNET_FORK_DENSE_LOSS=$v timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-mpjpe --no-roofline --no-bf16-legs --no-native-leg 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")" >> $out
done; done
cat $out
