"""GPU micro-benchmark of the implicit-GEMM entry points on the layer shapes of the B=32 256x256 train step.
Prints TFLOP/s (algorithmic 2*M*N*K) per shape and pass.  python tools/gemm_bench.py [filter]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pdfnet_amd import hip
from pdfnet_amd.hip import ptr, stream

L = hip.lib()
CL = torch.channels_last
B = 32
CONVS = [  # name, Cin, H, Cout, k, stride, pad
    ("conv1_7x7s2", 3, 256, 64, 7, 2, 3), ("l1.conv1_1x1", 256, 64, 64, 1, 1, 0), ("l1.conv2_3x3", 64, 64, 64, 3, 1, 1),
    ("l1.conv3_1x1", 64, 64, 256, 1, 1, 0), ("l2.conv2_3x3s2", 128, 64, 128, 3, 2, 1), ("l2.conv2_3x3", 128, 32, 128, 3, 1, 1),
    ("l2.conv1_1x1", 512, 32, 128, 1, 1, 0), ("l2.conv3_1x1", 128, 32, 512, 1, 1, 0), ("l3.conv2_3x3", 256, 16, 256, 3, 1, 1),
    ("l3.conv1_1x1", 1024, 16, 256, 1, 1, 0), ("l3.conv3_1x1", 256, 16, 1024, 1, 1, 0), ("l4.conv2_3x3", 512, 8, 512, 3, 1, 1),
    ("l4.conv3_1x1", 512, 8, 2048, 1, 1, 0), ("l4.down_1x1s2", 1024, 16, 2048, 1, 2, 0),
    ("p2/head_3x3", 256, 64, 256, 3, 1, 1), ("feat_3x3", 1024, 64, 256, 3, 1, 1), ("dec_3x3@64", 128, 64, 128, 3, 1, 1),
    ("head_1x1_122", 256, 64, 122, 1, 1, 0), ("dec_1x1", 2048, 8, 128, 1, 1, 0),
]
LINS = [("netR1.0", 1 << 20, 16, 64), ("netR1.3", 1 << 20, 64, 64), ("netR1.6", 1 << 20, 64, 128), ("netR2.0", 1 << 18, 144, 128),
        ("netR2.6", 1 << 18, 128, 256), ("netR3.3", 4096, 512, 512), ("gcn.fc1.l0", 2016, 1024, 256), ("gcn.fc.l1", 4032, 512, 128),
        ("gcn.fc.l2", 8064, 256, 64), ("attn.qkv.l2", 16128, 64, 64)]


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


BF16 = os.environ.get("PDF_BENCH_BF16", "0") != "0"      # bf16 kernels with bf16 shadows of every operand
WINO = os.environ.get("PDF_BENCH_WINOGRAD", "0") != "0"  # fp32: the `_x` forms with a Winograd workspace (what the train step calls)


def wino_opts(N, H, Cin, Cout, k, s, p, backward, dev):
    """(keep-alive tuple, byref(PdfCallOpts)) with the Winograd workspace of this pass, or (None, None) when the layer is not eligible."""
    import ctypes
    if not WINO or BF16:
        return None, None
    n = L.pdf_conv2d_winograd_workspace_floats(N, H, H, Cin, Cout, k, k, s, p, backward)
    if n <= 0:
        return None, None
    ws = torch.empty(n, device=dev)
    o = hip.CallOpts()
    o.ws, o.ws_floats = ws.data_ptr(), n
    return (ws, o), ctypes.byref(o)


def sh(*ts):
    """bf16 mode: hand the library the bf16 copies of the next call's two operands."""
    if BF16:
        L.pdf_set_bf16_operands(ptr(ts[0]), ptr(ts[1]))


def main():
    flt = sys.argv[1] if len(sys.argv) > 1 else ""
    dev = 'cuda'
    if BF16:
        L.pdf_set_gemm_precision(1)
    tot = {}
    for name, Cin, H, Cout, k, s, p in CONVS:
        if flt not in name:
            continue
        OH = (H + 2 * p - k) // s + 1
        x = torch.randn(B, Cin, H, H, device=dev).contiguous(memory_format=CL)
        w = (torch.randn(Cout, Cin, k, k, device=dev) * 0.05).contiguous(memory_format=CL)
        y = torch.empty(B, Cout, OH, OH, device=dev).contiguous(memory_format=CL)
        dy = torch.randn_like(y)
        dx = torch.zeros_like(x)
        dw = torch.empty_like(w)
        n = L.pdf_wgrad_workspace_floats(B * OH * OH, Cout, k * k * Cin)
        ws = torch.empty(max(n, 1), device=dev)
        fl = 2.0 * B * OH * OH * Cout * Cin * k * k
        x16, w16, dy16 = (t.to(torch.bfloat16) for t in (x, w, dy))
        k0, o0 = wino_opts(B, H, Cin, Cout, k, s, p, 0, dev)
        k1, o1 = wino_opts(B, H, Cin, Cout, k, s, p, 1, dev)
        k2, o2 = wino_opts(B, H, Cin, Cout, k, s, p, 2, dev)
        if WINO and not BF16:
            t_f = timeit(lambda: L.pdf_conv2d_fwd_x(ptr(x), ptr(w), None, ptr(y), B, H, H, Cin, Cin, Cout, k, k, s, p, OH, OH, Cout, 0, stream(), o0))
            t_d = timeit(lambda: L.pdf_conv2d_bwd_data_x(ptr(dy), ptr(w), ptr(dx), B, H, H, Cin, Cin, Cout, k, k, s, p, OH, OH, Cout, stream(), o1))
            t_w = timeit(lambda: L.pdf_conv2d_bwd_weight_x(ptr(x), ptr(dy), ptr(dw), None, ptr(ws), n, B, H, H, Cin, Cin, Cout, k, k, s, p, OH, OH, Cout, 0, stream(), o2))
            name = name + ("*" if o0 is not None else "")                 # * = Winograd path taken
        else:
            t_f = timeit(lambda: (sh(x16, w16), L.pdf_conv2d_fwd(ptr(x), ptr(w), None, ptr(y), B, H, H, Cin, Cin, Cout, k, k, s, p, OH, OH, Cout, 0, stream())))
            t_d = timeit(lambda: (sh(dy16, w16), L.pdf_conv2d_bwd_data(ptr(dy), ptr(w), ptr(dx), B, H, H, Cin, Cin, Cout, k, k, s, p, OH, OH, Cout, stream())))
            t_w = timeit(lambda: (sh(x16, dy16), L.pdf_conv2d_bwd_weight(ptr(x), ptr(dy), ptr(dw), None, ptr(ws), n, B, H, H, Cin, Cin, Cout, k, k, s, p, OH, OH, Cout, 0, stream())))
        print("%-16s M=%7d N=%5d K=%5d  %7.1f GF | fwd %6.3f ms %6.1f TF | bwd_data %6.3f ms %6.1f TF | bwd_w %6.3f ms %6.1f TF" %
              (name, B * OH * OH, Cout, Cin * k * k, fl / 1e9, t_f * 1e3, fl / t_f / 1e12, t_d * 1e3, fl / t_d / 1e12, t_w * 1e3, fl / t_w / 1e12), flush=True)
    for name, M, K, N in LINS:
        if flt not in name:
            continue
        x = torch.randn(M, K, device=dev)
        w = torch.randn(N, K, device=dev) * 0.05
        y = torch.empty(M, N, device=dev)
        dy = torch.randn(M, N, device=dev)
        dx = torch.empty(M, K, device=dev)
        dw = torch.empty(N, K, device=dev)
        n = L.pdf_wgrad_workspace_floats(M, N, K)
        ws = torch.empty(max(n, 1), device=dev)
        fl = 2.0 * M * N * K
        by = 4.0 * (M * K + M * N)
        t_f = timeit(lambda: L.pdf_linear_fwd(ptr(x), ptr(w), None, ptr(y), M, N, K, K, K, N, 1, stream()))
        t_d = timeit(lambda: L.pdf_linear_bwd_data(ptr(dy), ptr(w), ptr(dx), M, N, K, N, K, K, stream()))
        t_w = timeit(lambda: L.pdf_linear_bwd_weight(ptr(x), ptr(dy), ptr(dw), None, ptr(ws), n, M, N, K, K, N, 0, stream()))
        print("%-16s M=%7d N=%5d K=%5d  %7.1f GF | fwd %6.3f ms %6.1f TF %5.2f TB/s | bwd_data %6.3f ms %6.1f TF | bwd_w %6.3f ms %6.1f TF" %
              (name, M, N, K, fl / 1e9, t_f * 1e3, fl / t_f / 1e12, by / t_f / 1e12, t_d * 1e3, fl / t_d / 1e12, t_w * 1e3, fl / t_w / 1e12), flush=True)


if __name__ == "__main__":
    main()
