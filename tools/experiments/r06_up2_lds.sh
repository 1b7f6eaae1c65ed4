# pdf_upsample2x_bwd: the LDS-staged gather (default) against the plain gather kernel (PDF_UP2_BWD_LDS=0): parity, rates at the dense decoders' shapes
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
out=gpurun_out/r06_up2_lds.txt
: > $out
timeout 600 python -m pytest tests/test_ops_gpu.py -q -k "upsample" 2>&1 | tail -2 >> $out
for v in 0 1; do
PDF_UP2_BWD_LDS=$v timeout 300 python - >> $out 2>&1 <<PY
import torch
from pdfnet_amd import hip
from pdfnet_amd.hip import ptr, stream
L = hip.lib()
print("PDF_UP2_BWD_LDS=$v")
for (N, H, W, C) in ((32, 64, 64, 128), (32, 32, 32, 128), (32, 16, 16, 128), (32, 8, 8, 128)):
    dy = torch.randn(N, 2 * H, 2 * W, C, device='cuda')
    dx = torch.empty(N, H, W, C, device='cuda')
    for _ in range(3): L.pdf_upsample2x_bwd(ptr(dy), N, H, W, C, ptr(dx), stream())
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): L.pdf_upsample2x_bwd(ptr(dy), N, H, W, C, ptr(dx), stream())
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    mb = (dy.numel() + dx.numel()) * 4 / 1e6
    print("  %3d x %3d x %3d x %3d: %7.1f MB  %.4f ms  %7.1f GB/s" % (N, H, W, C, mb, ms, mb / ms))
PY
done
cat $out
