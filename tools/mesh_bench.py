"""Fused mesh decoder (csrc/meshdec.hip) against the unfused kernel chain, one DualGraphLayer at a time at the bench's batch (B = 32 per GPU):
forward and forward + backward, HIP events on the launch stream, side streams joined inside the timed region.
    python tools/mesh_bench.py [B]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from pdfnet_amd import functional as F
from pdfnet_amd.networks import intaghand_decoder as D

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
g = D.load_graph_constants()


def timed(fn, n=20, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


tot = {}
for level in (0, 1, 2):
    V, cin, cout = (63, 126, 252)[level], (512, 256, 128)[level], (256, 128, 64)[level]
    torch.manual_seed(0)
    layer = D.DualGraphLayer(V, cin, cout, g['ell_left'][level], g['ell_right'][level], 4, [12, 24, 48][level], 256, cout, 4, 0.05).cuda().train()
    x = torch.randn(2, B, V, cin, device='cuda')
    gy = torch.randn(2, B, V, cout, device='cuda')
    # FLOPs of the level's Linear layers (forward): rows x (2 K N) summed
    R = 2 * B * V
    lin = 2 * R * cout * (2 * cin + 2 * cout + cin) + 3 * 2 * R * cout * (2 * cout + 2 * cout + cout) + 2 * 6 * 2 * R * cout * cout
    att = 2 * 2 * B * 4 * 4 * V * V * (cout // 4)
    for fused in (False, True):
        F.MESH_FUSED = fused

        def fwd():
            with torch.no_grad():
                layer(x)

        def fwdbwd():
            xr = x.clone().requires_grad_()
            layer.zero_grad(set_to_none=True)
            layer(xr).backward(gy)
            F.join_wgrad()
        layer.eval()
        te = timed(fwd)
        layer.train()
        tf = timed(fwd)
        tb = timed(fwdbwd)
        tot.setdefault(fused, [0.0, 0.0, 0.0])
        for i, t in enumerate((te, tf, tb)):
            tot[fused][i] += t
        print("level %d (V %3d, C %3d) %-8s eval fwd %7.3f ms  train fwd %7.3f ms  fwd + bwd %7.3f ms   [fwd %.1f GFLOP linear + %.2f attention: %.1f TFLOP/s eval]"
              % (level, V, cout, "fused" if fused else "unfused", te, tf, tb, lin / 1e9, att / 1e9, (lin + att) / te / 1e9), flush=True)
F.MESH_FUSED = True
for fused in (False, True):
    print("%-8s all levels: eval fwd %.3f ms, train fwd %.3f ms, fwd + bwd %.3f ms" % ("fused" if fused else "unfused", *tot[fused]))
