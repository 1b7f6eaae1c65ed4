"""r04: Winograd F(4x4,3x3) weight gradient vs the direct weight-gradient kernels and vs float64."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import torch
import torch.nn.functional as TF
from pdfnet_amd import functional as F

def timeit(fn, iters=8):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters

for (N, Cin, H, Cout, bias) in ((8, 256, 64, 256, True), (8, 128, 64, 128, False), (32, 1024, 64, 256, False), (32, 256, 64, 256, True), (32, 128, 32, 128, True)):
    g = torch.Generator().manual_seed(1)
    x = torch.randn(N, Cin, H, H, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * (Cin * 9) ** -0.5
    b = torch.randn(Cout, generator=g) if bias else None
    gy = torch.randn(N, Cout, H, H, generator=g)
    res = {}
    for mode in (True, False):
        F.WINOGRAD = mode
        xd = x.cuda().contiguous(memory_format=torch.channels_last).requires_grad_()
        wd = w.cuda().contiguous(memory_format=torch.channels_last).requires_grad_()
        bd = b.cuda().requires_grad_() if bias else None
        out = F.conv2d(xd, wd, bd, 1, 1, 0)
        out.backward(gy.cuda())
        F.join_wgrad()
        torch.cuda.synchronize()
        def bw():
            wd.grad = None
            o = F.conv2d(xd.detach(), wd, None, 1, 1, 0)
            o.backward(gy.cuda()); F.join_wgrad()
        res[mode] = (wd.grad.detach().cpu().clone(), bd.grad.cpu().clone() if bias else None, timeit(bw))
    d = (res[True][0] - res[False][0]).abs().max()
    line = "N=%d %d->%d @%d: max|dW| %.1f  |wino - direct| %.2e" % (N, Cin, Cout, H, float(res[False][0].abs().max()), float(d))
    if bias:
        line += "  db diff %.2e" % float((res[True][1] - res[False][1]).abs().max())
    if N <= 8:
        xr, wr = x.double().requires_grad_(), w.double().requires_grad_()
        TF.conv2d(xr, wr, None, 1, 1).backward(gy.double())
        line += "  vs fp64: wino %.2e direct %.2e" % (float((res[True][0].double() - wr.grad).abs().max()), float((res[False][0].double() - wr.grad).abs().max()))
    line += "   fwd+bwd(w only) time: wino %.3f ms, direct %.3f ms" % (res[True][2], res[False][2])
    print(line)
