"""BatchNorm entry points through the C-ABI on the (C, R) pairs of the B=32 256x256 step: event time per call and algorithmic
GB/s (forward 3 passes, backward 5; +1 / +3 with a residual).  Usage: python tools/experiments/r04/bn_bench.py [filter]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import torch
from pdfnet_amd import hip
from pdfnet_amd.hip import ptr, stream

L = hip.lib()
dev = 'cuda'
SHAPES = [  # C, R, relu (0 none, 1 from y, 2 recomputed), residual
    (256, 524288, 2, 0), (128, 1 << 20, 0, 0), (512, 262144, 1, 1),       # 'cold' proxies of the 134 MB tensors: 537 MB each, nothing stays in the 256 MiB Infinity Cache
    (64, 1 << 20, 2, 0), (64, 524288, 2, 0), (64, 131072, 2, 0), (256, 131072, 1, 1), (256, 131072, 2, 0), (256, 131072, 0, 0),
    (128, 131072, 0, 0), (128, 131072, 2, 0), (128, 262144, 2, 0), (128, 32768, 2, 0), (512, 32768, 1, 1), (256, 8192, 2, 0),
    (1024, 8192, 1, 1), (512, 2048, 2, 0), (2048, 2048, 1, 1)]


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


def main():
    flt = sys.argv[1] if len(sys.argv) > 1 else ""
    tf = tb = bf = bb = 0.0
    for C, R, relu, res in SHAPES:
        name = "C=%d R=%d relu=%d res=%d" % (C, R, relu, res)
        if flt not in name:
            continue
        x, dy = torch.randn(R, C, device=dev), torch.randn(R, C, device=dev)
        r = torch.randn(R, C, device=dev) if res else None
        y, dx = torch.empty_like(x), torch.empty_like(x)
        dres = torch.empty_like(x) if res else None
        g, b = torch.ones(C, device=dev), torch.zeros(C, device=dev)
        rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
        mean, rstd, sc, sh, dg, db = (torch.empty(C, device=dev) for _ in range(6))
        n = L.pdf_bn_workspace_floats(C, R)
        ws = torch.empty(n + 3 * C, device=dev)
        fwd = lambda: L.pdf_bn_train_fwd(ptr(x), C, C, R, ptr(g), ptr(b), ptr(rm), ptr(rv), 0.1, 1e-5, ptr(r) if res else None, C, 1 if relu else 0,
                                         ptr(y), C, ptr(mean), ptr(rstd), ptr(sc), ptr(sh), ptr(ws), stream())
        bwd = lambda: L.pdf_bn_train_bwd(ptr(dy), C, ptr(y) if relu == 1 else None, C, relu, ptr(x), C, ptr(mean), ptr(rstd), ptr(g), ptr(sc), ptr(sh), C, R,
                                         ptr(dx), C, ptr(dres) if res else None, C, ptr(dg), ptr(db), 0, ptr(ws), stream())
        t_f, t_b = timeit(fwd), timeit(bwd)
        nb = R * C * 4
        nf, nbw = nb * (3 + (1 if res else 0)), nb * (5 + (3 if res else 0))
        print("%-34s fwd %8.1f us %6.2f TB/s | bwd %8.1f us %6.2f TB/s" % (name, t_f * 1e6, nf / t_f / 1e12, t_b * 1e6, nbw / t_b / 1e12), flush=True)
        tf += t_f; tb += t_b; bf += nf; bb += nbw
    print("all: fwd %.3f ms %.2f TB/s | bwd %.3f ms %.2f TB/s" % (tf * 1e3, bf / tf / 1e12, tb * 1e3, bb / tb / 1e12))


if __name__ == "__main__":
    main()
