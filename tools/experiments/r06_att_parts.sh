# workgroups per attention-backward pass of the mesh decoder (PDF_MESH_ATT_PARTS, digits = level 0 / 1 / 2; default 2-2-4) re-checked with eight waves per workgroup
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
out=gpurun_out/r06_att_parts.txt
: > $out
run() { timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-mpjpe --no-roofline --no-bf16-legs --no-native-leg 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
for r in 1 2; do
  for v in 224 111 124 248 228; do echo "round $r PDF_MESH_ATT_PARTS=$v: $(PDF_MESH_ATT_PARTS=$v run)" >> $out; done
done
cat $out
