# the fp32 step at batch sizes other than the headline's: every entry point (x3 / Winograd / split-K choices depend on the batch) must take them
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
out=gpurun_out/r06_batch_sweep.txt
: > $out
for b in 2 3 5 16 24 48 64 96; do
  echo "B=$b" >> $out
  timeout 400 python bench.py --batch $b --steps 6 --warmup 3 --no-cpu-baseline --no-mpjpe --no-roofline --no-bf16-legs --no-native-leg --no-collective-path 2>&1 | tail -1 | python -c "
import sys, json
s = sys.stdin.read()
try:
    d = json.loads(s); print(d['value'], d['ms_per_step'], 'final loss', d['config'].get('final_loss'))
except Exception as e:
    print('FAILED', s[-400:])
" >> $out 2>&1
done
cat $out
