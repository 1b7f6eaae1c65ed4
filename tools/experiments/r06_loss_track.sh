# training dynamics with x3 on / off: the same fixed synthetic batch, same seeds, 150 Adam steps -- the final losses must agree closely (x3 differs from the
# native fp32 MFMA at the 1e-7 level per product) and both must have fallen
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
out=gpurun_out/r06_loss_track.txt
: > $out
for s in 5 150; do for x in 1 0; do
  echo "PDF_X3=$x steps=$s: $(PDF_X3=$x timeout 600 python bench.py --steps $s --warmup 1 --no-cpu-baseline --no-mpjpe --no-roofline --no-bf16-legs --no-native-leg --no-collective-path 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('final_loss', d['config']['final_loss'], 'img/s', d['value'])")" >> $out
done; done
cat $out
