"""Microbenchmark of pdf_layernorm_fused_bwd at the mesh decoder's shapes (rows = 2 hands x B x V)."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pdfnet_amd import hip
from pdfnet_amd.hip import ptr, stream
L = hip.lib()
dev = torch.device('cuda')
for (R, F) in ((16128, 64), (8064, 128), (4032, 256), (4032, 512), (2016, 512), (2016, 1024)):
    dy = torch.randn(R, F, device=dev); y = torch.randn(R, F, device=dev); z = torch.randn(R, F, device=dev)
    g0 = torch.randn(F, device=dev); g1 = torch.randn(F, device=dev)
    mean = torch.randn(R, device=dev); rstd = torch.rand(R, device=dev) + 0.5
    dz = torch.empty(R, F, device=dev)
    dg0 = torch.zeros(F, device=dev); db0 = torch.zeros(F, device=dev); dg1 = torch.zeros(F, device=dev); db1 = torch.zeros(F, device=dev)
    mode = os.environ.get('LN_MODE', 'both')
    def run():
        if mode == 'data':
            L.pdf_layernorm_fused_bwd(ptr(dy), F, ptr(y), F, 1, ptr(z), F, F, R, R // 2, ptr(g0), ptr(g1), ptr(mean), ptr(rstd), None, 0,
                                      ptr(dz), F, None, 0, ctypes.c_float(0.0), 0, None, None, None, None, None, stream())
            return
        if mode == 'params':
            L.pdf_layernorm_fused_bwd(ptr(dy), F, ptr(y), F, 1, ptr(z), F, F, R, R // 2, ptr(g0), ptr(g1), ptr(mean), ptr(rstd), None, 0,
                                      None, F, None, 0, ctypes.c_float(0.0), 0, None, ptr(dg0), ptr(db0), ptr(dg1), ptr(db1), stream())
            return
        L.pdf_layernorm_fused_bwd(ptr(dy), F, ptr(y), F, 1, ptr(z), F, F, R, R // 2, ptr(g0), ptr(g1), ptr(mean), ptr(rstd), None, 0,
                                  ptr(dz), F, None, 0, ctypes.c_float(0.0), 0, None, ptr(dg0), ptr(db0), ptr(dg1), ptr(db1), stream())
    for _ in range(10): run()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(200): run()
    b.record(); torch.cuda.synchronize()
    print("R %6d F %5d  %.1f us" % (R, F, a.elapsed_time(b) * 5))
