cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3aa
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q 2>&1 | tail -1
for lib in libpdfnet_hip_head.so libpdfnet_hip.so; do
 for f in feat p2/head dec_3x3 l1.conv2 l2.conv2_3x3 netR1.6; do
  echo "== LIB $lib $f"; PDFNET_HIP_LIB=$GRAFT_REPO_ROOT/pdfnet_amd/$lib timeout 120 python tools/gemm_bench.py $f 2>&1 | grep "fwd" | sed -e 's/ GF |/ |/' | cut -c1-125
 done; done
for i in 1 2; do
PDFNET_HIP_LIB=$GRAFT_REPO_ROOT/pdfnet_amd/libpdfnet_hip_head.so timeout 300 python bench.py --steps 20 --warmup 8 --no-cpu-baseline --no-roofline --no-mpjpe --no-bf16-legs > gpurun_out/r3aa/b_old$i.json 2>/dev/null
timeout 300 python bench.py --steps 20 --warmup 8 --no-cpu-baseline --no-roofline --no-mpjpe --no-bf16-legs > gpurun_out/r3aa/b_new$i.json 2>/dev/null
done
for f in gpurun_out/r3aa/b_*.json; do echo $f; python -c "
import json,sys
d=json.loads(open('$f').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; done
