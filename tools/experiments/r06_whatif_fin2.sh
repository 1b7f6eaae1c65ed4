# what-if: every BatchNorm finalize launch (forward tile form, backward) issued TWICE -- the added time per step = the cost of those 138 launches on the
# dependent chain (their execution + dispatch latency).  (pdfnet_amd/libpdfnet_hip_whatif.so, built from a temporary patch of norm.hip that is not in the tree)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
out=gpurun_out/r06_whatif_fin2.txt
: > $out
export PDFNET_HIP_LIB=$GRAFT_REPO_ROOT/pdfnet_amd/libpdfnet_hip_whatif.so
for r in 1 2; do for w in 0 1; do
  echo "round $r finalize launches doubled=$w: img/s, ms/step" >> $out
  if [ $w = 1 ]; then export PDF_WHATIF_FIN2=1; else unset PDF_WHATIF_FIN2; fi
  timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-mpjpe --no-roofline --no-bf16-legs --no-native-leg 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" >> $out 2>&1
done; done
cat $out
