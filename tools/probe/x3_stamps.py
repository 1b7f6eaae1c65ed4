"""Phase clocks of x3gemm_nt (diagnostic build of gemm_x3.hip, -DX3_STAMPS=1): where wave 0 of block 0 spends its cycles per K-step, and the
clock the chip holds inside the K loop.   PDFNET_HIP_LIB=tools/probe/libpdfnet_hip_stamps.so python tools/probe/x3_stamps.py"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pdfnet_amd import hip
from pdfnet_amd.hip import ptr, stream

L = hip.lib()
dev = "cuda"
for name, P, M, N, K in (("feat.fwd", 36, 8192, 256, 1024), ("head.fwd", 36, 8192, 256, 256), ("p5-like", 1, 2048, 16384, 2048)):
    A = torch.randn(P, M, K, device=dev)
    B = torch.randn(P, N, K, device=dev) * 0.05
    C = torch.empty(P, M, N, device=dev)
    A3 = torch.empty((3,) + tuple(A.shape), dtype=torch.bfloat16, device=dev)
    B3 = torch.empty((3,) + tuple(B.shape), dtype=torch.bfloat16, device=dev)
    L.pdf_x3_split(ptr(A), ptr(A3), A.numel(), A.numel(), stream())
    L.pdf_x3_split(ptr(B), ptr(B3), B.numel(), B.numel(), stream())
    for v in (0, 1):
        for _ in range(3):                                  # back-to-back launches: the clock settles
            for _ in range(20):
                L.pdf_x3_batched_gemm_nt(ptr(A3), A.numel(), ptr(B3), B.numel(), ptr(C), P, M * K, N * K, M * N, M, N, K, v, 6, stream())
            torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            L.pdf_x3_batched_gemm_nt(ptr(A3), A.numel(), ptr(B3), B.numel(), ptr(C), P, M * K, N * K, M * N, M, N, K, v, 6, stream())
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        buf = (ctypes.c_ulonglong * 16)()
        n = L.pdf_debug_x3_stamps(buf)
        tw, tr, tm, tot, rt, nk = [buf[i] for i in range(6)]
        nk = max(nk, 1)
        mf = 48 * 32                                        # MFMA cycles per K-step and wave (both variants: 2 x 2 tiles x 2 sub-steps x 6 products)
        print("%-9s v%d  %.3f ms  %.1f TF-eq | per K-step (cycles): wait+barrier %5.0f  reads %5.0f  mfma/dma slots %5.0f  (sum %5.0f; MFMA issue time %d) | "
              "clock %.2f GHz (s_memtime / s_memrealtime at 100 MHz)" % (name, v, ms, 2.0 * P * M * N * K / ms * 1e-9, tw / nk, tr / nk, tm / nk, tot / nk, mf,
                                                                      tot / max(rt, 1) * 0.1))
