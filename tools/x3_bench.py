"""x3 arithmetic (csrc/gemm_x3.hip) beside the native fp32-MFMA batched product: error against float64 and time, on the Winograd-domain
product shapes of the B=32 256x256 train step.   python tools/x3_bench.py [filter] [--variants 0,1] [--nprod 6,9]

The gate: ship x3 only where its max error against float64 is <= the native kernel's on the same operands and its
TFLOP/s-equivalent (2 M N K per product / time) is >= 1.3x the native kernel's."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pdfnet_amd import hip
from pdfnet_amd.hip import ptr, stream

L = hip.lib()
# name, batch (planes), M (tiles), N, K        F(4x4) at B = 32: T = 32 (H/4) (W/4)
NT = [
    ("feat.fwd 1024->256@64", 36, 8192, 256, 1024), ("feat.bwd 256->1024@64", 36, 8192, 1024, 256), ("head.fwd 256->256@64", 36, 8192, 256, 256),
    ("dec.fwd 128->128@64", 36, 8192, 128, 128), ("l2.fwd 128->128@32", 36, 2048, 128, 128), ("l3.fwd 256->256@16", 36, 512, 256, 256),
]
# name, batch, M (tiles: the reduction), NI (Cout), NJ (Cin), splits
TN = [
    ("feat.wgrad", 36, 8192, 256, 1024, 8), ("head.wgrad", 36, 8192, 256, 256, 16), ("dec.wgrad", 36, 8192, 128, 128, 32),
    ("l2.wgrad", 36, 2048, 128, 128, 8), ("l3.wgrad", 36, 512, 256, 256, 2),
]


def timeit(fn, iters=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def split3(x):
    """fp32 tensor -> bf16 tensor [3, *x.shape] of x3 planes (the library's kernel)."""
    o = torch.empty((3,) + tuple(x.shape), dtype=torch.bfloat16, device=x.device)
    L.pdf_x3_split(ptr(x), ptr(o), x.numel(), x.numel(), stream())
    return o


def errs(c, ref):
    d = (c.double() - ref).abs()
    return d.max().item() / ref.abs().max().item(), (d.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item()


def wino_like(shape, dev, scale):
    """Operands with the spread of transform-domain values: N(0, 1) times a per-plane factor over two decades."""
    x = torch.randn(shape, device=dev)
    f = torch.logspace(-1, 1, shape[0], device=dev).view(-1, *([1] * (len(shape) - 1)))
    return (x * f * scale).contiguous()


def deconvs():
    """The pyramid's transposed convolutions through the Python op (forward, backward-data + weight gradient), x3 form off / on."""
    from pdfnet_amd import functional as F
    for name, Cin, H, Cout, k, s, p in (("p5 2048->256 k8s8 @8", 2048, 8, 256, 8, 8, 0), ("p4 1024->256 k4s4 @16", 1024, 16, 256, 4, 4, 0),
                                        ("p3 512->256 k4s2p1 @32", 512, 32, 256, 4, 2, 1)):
        x = torch.randn(32, Cin, H, H, device="cuda").contiguous(memory_format=torch.channels_last)
        w = (torch.randn(Cin, Cout, k, k, device="cuda") * Cin ** -0.5).contiguous(memory_format=torch.channels_last)
        b = torch.randn(Cout, device="cuda")
        fl = 2.0 * 32 * H * H * Cin * k * k * Cout
        outs = {}
        gy = None
        for x3 in (False, True):
            F.X3_DECONV = x3
            xr = x.clone().requires_grad_()
            with torch.no_grad():
                tf = timeit(lambda: F.deconv2d(xr, w, b, s, p))
            y = F.deconv2d(xr, w, b, s, p)
            gy = torch.randn_like(y) if gy is None else gy
            tb = timeit(lambda: torch.autograd.grad(y, xr, gy, retain_graph=True))
            outs[x3] = (y.detach().clone(), torch.autograd.grad(y, xr, gy, retain_graph=True)[0].clone())
            wr = w.clone().requires_grad_()

            def wgrad():
                yy = F.deconv2d(x, wr, None, s, p)
                torch.autograd.grad(yy, wr, gy)
                F.join_wgrad()
            tw = timeit(wgrad) - tf                             # (forward + weight gradient, minus the forward)
            print("%-24s x3 %-5s  forward %.3f ms (%.1f TF-eq)   backward-data %.3f ms (%.1f TF-eq)   weight gradient ~%.3f ms (%.1f TF-eq)"
                  % (name, x3, tf * 1e3, fl / tf * 1e-12, tb * 1e3, fl / tb * 1e-12, tw * 1e3, fl / tw * 1e-12))
        dy = (outs[True][0] - outs[False][0]).abs().max().item() / outs[False][0].abs().max().item()
        dx = (outs[True][1] - outs[False][1]).abs().max().item() / outs[False][1].abs().max().item()
        print("%-24s x3 vs native: max |dy| / max|y| = %.2e, max |d dx| / max|dx| = %.2e" % (name, dy, dx))
    F.X3_DECONV = True


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    flt = args[0] if args else ""
    if flt == "deconv":
        return deconvs()
    opt = {a.split("=")[0]: a.split("=")[1] for a in sys.argv[1:] if a.startswith("--") and "=" in a}
    variants = [int(v) for v in opt.get("--variants", "0,1").split(",")]
    nprods = [int(v) for v in opt.get("--nprod", "6").split(",")]
    dev = "cuda"
    torch.manual_seed(0)
    # the split itself: h + m + l == x exactly
    x = wino_like((4, 1024, 512), dev, 1.0)
    s3 = split3(x)
    back = s3[0].float() + s3[1].float() + s3[2].float()
    print("split: max |h + m + l - x| = %.3g (must be 0); planes' max |m| / |h| = %.2e, |l| / |h| = %.2e"
          % ((back - x).abs().max().item(), (s3[1].float().abs().max() / s3[0].float().abs().max()).item(),
             (s3[2].float().abs().max() / s3[0].float().abs().max()).item()))
    print("%-26s %5s %-9s %9s %9s %9s %9s %9s" % ("shape", "arm", "", "ms", "TF-eq", "max-err", "rms-err", "vs native"))
    for name, P, M, N, K in NT:
        if flt not in name:
            continue
        A = wino_like((P, M, K), dev, 1.0)
        B = wino_like((P, N, K), dev, 0.05)
        nref = 2
        ref = torch.bmm(A[:nref].double(), B[:nref].double().transpose(1, 2))
        C = torch.empty(P, M, N, device=dev)
        fl = 2.0 * P * M * N * K
        t_nat = timeit(lambda: L.pdf_batched_gemm_nt(ptr(A), ptr(B), ptr(C), P, M * K, N * K, M * N, M, N, K, stream()))
        e_nat = errs(C[:nref], ref)
        print("%-26s %5s %-9s %9.3f %9.1f %9.2e %9.2e" % (name, "fp32", "native", t_nat * 1e3, fl / t_nat * 1e-12, e_nat[0], e_nat[1]))
        A3, B3 = split3(A), split3(B)
        for v in variants:
            for npd in nprods:
                C.zero_()
                try:
                    t = timeit(lambda: L.pdf_x3_batched_gemm_nt(ptr(A3), A.numel(), ptr(B3), B.numel(), ptr(C), P, M * K, N * K, M * N, M, N, K, v, npd, stream()))
                except RuntimeError as e:
                    print("%-26s %5s v%d n%d  %s" % (name, "x3", v, npd, e))
                    continue
                e = errs(C[:nref], ref)
                print("%-26s %5s v%d n%d     %9.3f %9.1f %9.2e %9.2e %8.2fx  err %.2fx" % (name, "x3", v, npd, t * 1e3, fl / t * 1e-12, e[0], e[1], t_nat / t, e[1] / e_nat[1]))
        del A, B, C, A3, B3, ref
    if not hasattr(L.cdll, "pdf_x3_batched_gemm_tn"):
        return
    for name, P, M, NI, NJ, splits in TN:
        if flt not in name:
            continue
        Pm = wino_like((P, M, NI), dev, 0.05)
        Q = wino_like((P, M, NJ), dev, 1.0)
        nref = 2
        ref = torch.bmm(Pm[:nref].double().transpose(1, 2), Q[:nref].double())
        fl = 2.0 * P * M * NI * NJ
        slab = torch.empty(P, splits, NI, NJ, device=dev)
        def used_splits(q):                               # rows per split rounded up to q, as the library plans it
            rps = -(-(-(-M // splits)) // q) * q
            return -(-M // rps)
        un = used_splits(16)
        t_nat = timeit(lambda: L.pdf_batched_gemm_tn(ptr(Pm), ptr(Q), ptr(slab), P, M * NI, M * NJ, M, NI, NJ, splits, stream()))
        c = slab.flatten()[:P * un * NI * NJ].view(P, un, NI, NJ)[:nref].sum(1)
        e_nat = errs(c, ref)
        print("%-26s %5s %-9s %9.3f %9.1f %9.2e %9.2e" % (name, "fp32", "native", t_nat * 1e3, fl / t_nat * 1e-12, e_nat[0], e_nat[1]))
        P3, Q3 = split3(Pm), split3(Q)
        for v in variants:
            for npd in nprods:
                slab.zero_()
                ux = used_splits(32)
                try:
                    t = timeit(lambda: L.pdf_x3_batched_gemm_tn(ptr(P3), Pm.numel(), ptr(Q3), Q.numel(), ptr(slab), P, M * NI, M * NJ, M, NI, NJ, splits, v, npd, stream()))
                except RuntimeError as e:
                    print("%-26s %5s v%d n%d  %s" % (name, "x3", v, npd, e))
                    continue
                c = slab.flatten()[:P * ux * NI * NJ].view(P, ux, NI, NJ)[:nref].sum(1)
                e = errs(c, ref)
                print("%-26s %5s v%d n%d     %9.3f %9.1f %9.2e %9.2e %8.2fx  err %.2fx" % (name, "x3", v, npd, t * 1e3, fl / t * 1e-12, e[0], e[1], t_nat / t, e[1] / e_nat[1]))
        del Pm, Q, slab, P3, Q3, ref


if __name__ == "__main__":
    main()
