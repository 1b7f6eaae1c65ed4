# tile / split-K rules of the native implicit-GEMM dispatch, re-checked inside today's step (they were set in rounds 2-3): one call, two rounds
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
out=gpurun_out/r06_ig_rules.txt
: > $out
run() { timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-mpjpe --no-roofline --no-bf16-legs --no-native-leg 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
for r in 1 2; do
  echo "round $r default: $(run)" >> $out
  for v in 400 900; do echo "round $r PDF_IG_T128=$v: $(PDF_IG_T128=$v run)" >> $out; done
  echo "round $r PDF_IG_SHORTK=0: $(PDF_IG_SHORTK=0 run)" >> $out
  for v in 384 768; do echo "round $r PDF_IG_SPLITK_TARGET=$v: $(PDF_IG_SPLITK_TARGET=$v run)" >> $out; done
  for v in 128 384; do echo "round $r PDF_IG_SPLITK_MAXT=$v: $(PDF_IG_SPLITK_MAXT=$v run)" >> $out; done
  echo "round $r PDF_IG_T32=0: $(PDF_IG_T32=0 run)" >> $out
done
cat $out
