"""Achieved HBM bandwidth of the streaming (non-GEMM) kernels on the shapes the B=32 256x256 step runs them at:
algorithmic bytes (each operand read / written once) / event time.  Usage: python tools/hbm_bench.py [filter]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pdfnet_amd import functional as F

dev = 'cuda'
B = 32
CL = torch.channels_last


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / iters


def report(name, nbytes, t):
    print("%-34s %8.1f MB  %8.3f ms  %6.2f TB/s" % (name, nbytes / 1e6, t * 1e3, nbytes / t / 1e12), flush=True)


def main():
    flt = sys.argv[1] if len(sys.argv) > 1 else ""
    # BatchNorm (train) + ReLU [+ residual]: (C, H) of the ResNet-50 trunk at 256x256 and the decoders
    for name, C, H, res in (("bn conv1 64@128", 64, 128, False), ("bn l1 64@64", 64, 64, False), ("bn l1 256@64 +res", 256, 64, True),
                            ("bn l2 128@32", 128, 32, False), ("bn l2 512@32 +res", 512, 32, True), ("bn l3 256@16", 256, 16, False),
                            ("bn l3 1024@16 +res", 1024, 16, True), ("bn l4 512@8", 512, 8, False), ("bn l4 2048@8 +res", 2048, 8, True),
                            ("bn feat 256@64", 256, 64, False), ("bn dec 128@128", 128, 128, False)):
        if flt not in name:
            continue
        x = torch.randn(B, C, H, H, device=dev).contiguous(memory_format=CL).requires_grad_()
        r = torch.randn(B, C, H, H, device=dev).contiguous(memory_format=CL).requires_grad_() if res else None
        g, b = torch.ones(C, device=dev).requires_grad_(), torch.zeros(C, device=dev).requires_grad_()
        rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
        n = x.numel() * 4
        y = F.batch_norm(x, g, b, rm, rv, True, 0.1, 1e-5, True, r)
        gy = torch.randn_like(y)
        t_f = timeit(lambda: F.batch_norm(x, g, b, rm, rv, True, 0.1, 1e-5, True, r))
        t_b = timeit(lambda: torch.autograd.grad(y, [x] + ([r] if res else []), gy, retain_graph=True))
        report(name + " fwd", n * (3 + (1 if res else 0)), t_f)               # stats read, apply read (+res), write
        report(name + " bwd", n * (8 if res else 5), t_b)                     # residual: (dy,x,y) twice + dx + dres; else (dy,x) twice + dx
    for name, C, H in (("l2norm 256@64", 256, 64),):
        if flt not in name:
            continue
        x = torch.randn(B, C, H, H, device=dev).contiguous(memory_format=CL).requires_grad_()
        w = torch.ones(C, device=dev).requires_grad_()
        y = F.l2norm(x, w)
        gy = torch.randn_like(y)
        n = x.numel() * 4
        report(name + " fwd", 2 * n, timeit(lambda: F.l2norm(x, w)))
        report(name + " bwd", 3 * n, timeit(lambda: torch.autograd.grad(y, [x, w], gy, retain_graph=True)))
    for name, C, H in (("up2 128@64", 128, 64), ("up2 64@128", 64, 128)):
        if flt not in name:
            continue
        x = torch.randn(B, C, H, H, device=dev).contiguous(memory_format=CL).requires_grad_()
        y = F.upsample2x(x)
        gy = torch.randn_like(y)
        n = x.numel() * 4
        report(name + " fwd", 5 * n, timeit(lambda: F.upsample2x(x)))
        report(name + " bwd", 5 * n, timeit(lambda: torch.autograd.grad(y, [x], gy, retain_graph=True)))
    if flt in "maxpool":
        x = torch.randn(B, 64, 128, 128, device=dev).contiguous(memory_format=CL).requires_grad_()
        y = F.maxpool3s2(x)
        gy = torch.randn_like(y)
        n = x.numel() * 4
        report("maxpool 64@128 fwd", n * 1.25 + n / 16, timeit(lambda: F.maxpool3s2(x)))
        report("maxpool 64@128 bwd", n * 1.25 + n / 16, timeit(lambda: torch.autograd.grad(y, [x], gy, retain_graph=True)))
    if flt in "knn group":
        # PointNet++ grouping, level 1 (1024 points -> 512 centroids x 64 neighbours) and level 2 (512 -> 128 x 64, 131(+pad) ch)
        for name, N, S, C, Cp in (("knn+group L1", 1024, 512, 3, 16), ("knn+group L2", 512, 128, 131, 144)):
            pts = torch.randn(2 * B, N, Cp, device=dev)
            pts[..., C:] = 0
            t = timeit(lambda: F.knn_ball_group(pts, C, S, 64, 0.1 if C == 3 else 0.3, Cp))
            report(name, 2 * B * (N * Cp * 4 + S * 64 * 4 + S * 64 * Cp * 4), t)
    if flt in "fps":
        x = torch.rand(2 * B, 4096, 3, device=dev)
        for S in (512, 1024):
            t = timeit(lambda: F.fps(x, S), iters=5)
            print("%-34s %8.3f ms  (%d clouds x 4096 points, %.2f us per pick: latency-bound)" % ("fps 4096 -> %d" % S, t * 1e3, 2 * B, t / S * 1e6))
    if flt in "adam":
        n = 100_500_000
        p, g, m, v = (torch.zeros(n, device=dev) for _ in range(4))
        corr = torch.tensor([0.1, 0.001], device=dev)
        from pdfnet_amd import hip
        L = hip.lib()
        t = timeit(lambda: L.pdf_adam_step(hip.ptr(p), hip.ptr(g), hip.ptr(m), hip.ptr(v), n, 1e-4, 0.9, 0.999, 1e-8, hip.ptr(corr), 1.0, hip.stream()))
        report("adam 100.5M", n * 4 * 7, t)


if __name__ == "__main__":
    main()
