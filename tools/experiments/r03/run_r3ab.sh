cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3ab
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_bf16_gpu.py -x -q 2>&1 | tail -1
for lib in libpdfnet_hip_head.so libpdfnet_hip.so; do
 for f in l2.conv l3.conv l4.conv netR2 netR3; do
  echo "== LIB $lib $f"; PDFNET_HIP_LIB=$GRAFT_REPO_ROOT/pdfnet_amd/$lib timeout 120 python tools/gemm_bench.py $f 2>&1 | grep "fwd" | sed -e 's/ GF |/ |/' | cut -c1-160
 done; done
for i in 1 2; do
PDFNET_HIP_LIB=$GRAFT_REPO_ROOT/pdfnet_amd/libpdfnet_hip_head.so timeout 300 python bench.py --steps 20 --warmup 8 --no-cpu-baseline --no-roofline --no-mpjpe --no-bf16-legs > gpurun_out/r3ab/b_old$i.json 2>/dev/null
timeout 300 python bench.py --steps 20 --warmup 8 --no-cpu-baseline --no-roofline --no-mpjpe --no-bf16-legs > gpurun_out/r3ab/b_new$i.json 2>/dev/null
done
for f in gpurun_out/r3ab/b_*.json; do echo $f; python -c "
import json,sys
d=json.loads(open('$f').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; done
