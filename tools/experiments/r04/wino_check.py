"""r04: Winograd F(2x2,3x3) path vs the direct kernels: values (against fp64 CPU) and time, per layer and pass."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import torch
import torch.nn.functional as TF
from pdfnet_amd import functional as F

def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters

for (N, Cin, H, Cout) in ((4, 256, 64, 256), (32, 1024, 64, 256), (32, 256, 64, 256), (32, 128, 64, 128)):
    g = torch.Generator().manual_seed(1)
    x = torch.randn(N, Cin, H, H, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * (Cin * 9) ** -0.5
    b = torch.randn(Cout, generator=g)
    gy = torch.randn(N, Cout, H, H, generator=g)
    res = {}
    for mode in (True, False):
        F.WINOGRAD = mode
        xd = x.cuda().contiguous(memory_format=torch.channels_last).requires_grad_()
        wd = w.cuda().contiguous(memory_format=torch.channels_last).requires_grad_()
        bd = b.cuda().requires_grad_()
        out = F.conv2d(xd, wd, bd, 1, 1, 1)
        out.backward(gy.cuda())
        F.join_wgrad()
        torch.cuda.synchronize()
        with torch.no_grad():
            tf = timeit(lambda: F.conv2d(xd, wd, bd, 1, 1, 1))
        res[mode] = (out.detach().cpu(), xd.grad.cpu(), tf)
    if N <= 4:
        xr, wr, br = x.double().requires_grad_(), w.double().requires_grad_(), b.double().requires_grad_()
        ref = TF.relu(TF.conv2d(xr, wr, br, 1, 1)); ref.backward(gy.double())
        for mode in (True, False):
            print("  %s: fwd err %.2e  dx err %.2e (max|ref| %.2f / %.2f)" % ("winograd" if mode else "direct  ", float((res[mode][0].double() - ref).abs().max()),
                  float((res[mode][1].double() - xr.grad).abs().max()), float(ref.abs().max()), float(xr.grad.abs().max())))
    fl = 2.0 * N * H * H * Cout * Cin * 9
    print("N=%d %d->%d @%d: fwd winograd %.3f ms (%.0f TF algorithmic) direct %.3f ms (%.0f TF);  max |wino - direct| fwd %.2e dx %.2e" % (
        N, Cin, Cout, H, res[True][2], fl / res[True][2] / 1e9, res[False][2], fl / res[False][2] / 1e9,
        float((res[True][0] - res[False][0]).abs().max()), float((res[True][1] - res[False][1]).abs().max())))
