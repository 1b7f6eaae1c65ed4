# Winograd for ResNet layer 4 (8x8 maps, 512 channels: F(2x2) 16 planes x 512 tiles = 8,192 < the 16,384 gate of PDF_WINOGRAD_MINPT): re-checked in today's step
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
out=gpurun_out/r06_wino_minpt.txt
: > $out
run() { timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-mpjpe --no-roofline --no-bf16-legs --no-native-leg 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
for r in 1 2; do
  echo "round $r default (16384): $(run)" >> $out
  for v in 8192 4096; do echo "round $r PDF_WINOGRAD_MINPT=$v: $(PDF_WINOGRAD_MINPT=$v run)" >> $out; done
done
cat $out
