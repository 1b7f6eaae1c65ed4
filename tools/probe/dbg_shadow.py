import torch, sys
sys.path.insert(0, '/root/repo')
from pdfnet_amd import functional as F, hip
F.set_gemm_precision('bf16')
L = hip.lib()
for cfg in [(2, 72, 9, 9, 40, 3, 1, 1), (2, 64, 16, 16, 128, 3, 1, 1)]:
    N, Cin, H, W, Cout, k, st, pad = cfg
    x = torch.randn(N, Cin, H, W).cuda().contiguous(memory_format=torch.channels_last).requires_grad_()
    w = torch.randn(Cout, Cin, k, k).cuda().contiguous(memory_format=torch.channels_last).requires_grad_()
    OH, OW = (H + 2 * pad - k) // st + 1, (W + 2 * pad - k) // st + 1
    dy = torch.randn(N, Cout, OH, OW).cuda().contiguous(memory_format=torch.channels_last)
    for t in (x, w, dy):
        F.attach_shadow(t, t.detach().to(torch.bfloat16))
    c0 = L.pdf_debug_shadow_operands()
    y = F.conv2d(x, w, None, st, pad, F.ACT_NONE)
    c1 = L.pdf_debug_shadow_operands()
    orig = F._Conv2d.backward
    def bw(ctx, dyy, dskip=None):
        print('  dy is same object', dyy is dy, 'shadow', F.shadow_of(dyy) is not None, 'ctx.s16', [s is not None for s in ctx.s16])
        return orig(ctx, dyy, dskip)
    F._Conv2d.backward = staticmethod(bw)
    y.backward(dy)
    F._Conv2d.backward = orig
    c2 = L.pdf_debug_shadow_operands()
    print(cfg, 'fwd used', c1 - c0, 'bwd used', c2 - c1)
