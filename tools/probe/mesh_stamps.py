"""Stage clocks of the fused mesh decoder's kernels (diagnostic build of csrc/meshdec.hip with -DMD_STAMPS=1: workgroup 0's thread 0 stamps the
shader clock at the stage boundaries).   PDFNET_HIP_LIB=<stamps build> python tools/probe/mesh_stamps.py [B]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pdfnet_amd import hip, functional as F
from pdfnet_amd.networks import intaghand_decoder as D

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
L = hip.lib()
g = D.load_graph_constants()
GCN = ["start", "fc1 + shortcut products (loads, L x)", "epilogue", "LN2", "L h", "fc2 product", "z epilogue", "LN3"]
for level in range(3):
    V, cin, cout = (63, 126, 252)[level], (512, 256, 128)[level], (256, 128, 64)[level]
    torch.manual_seed(0)
    layer = D.DualGraphLayer(V, cin, cout, g['ell_left'][level], g['ell_right'][level], 4, [12, 24, 48][level], 256, (256, 128, 64)[level], 4, 0.05).cuda().train()
    x = torch.randn(2, B, V, cin, device='cuda', requires_grad=True)
    for _ in range(3):
        out = layer(x)
        out.backward(torch.randn_like(out))
        F.join_wgrad()
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * (8 * 3 * 64))()
    if not L.cdll.pdf_debug_mesh_stamps(buf):
        print("this library was built without -DMD_STAMPS=1")
        sys.exit(1)
    st = lambda kid, n: buf[(kid * 3 + level) * 64 + n]
    print("level %d (V %d, C %d), B %d: shader clocks (100 MHz s_memtime-free counter; ~2 GHz) of workgroup 0" % (level, V, cout, B))
    # forward GCN kernel: stamp 0 at entry, per block 1..8, 40 at the end (LN + q / k / v of the self attention)
    print("  mesh_gcn_kernel: entry -> first block %d; total %d" % (st(0, 1) - st(0, 0), st(0, 40) - st(0, 0)))
    for blk in range(4):
        d = [st(0, k + 1 + blk * 8) - st(0, k + blk * 8) for k in range(1, 8)]
        print("    block %d: %s" % (blk, ", ".join("%s %d" % (n, v) for n, v in zip(GCN[1:], d))))
    print("    tail (LN + q / k / v): %d" % (st(0, 40) - st(0, 8 + 3 * 8)))
    print("  mesh_att_kernel (last launched = cross): K, V loads %d, attention %d, fc + z epilogue %d, LN %d, f1 / f2 + epilogues %d" %
          tuple(st(1, k + 1) - st(1, k) for k in range(5)))
    print("  mesh_gcn_bwd_kernel block 3..1: " + "; ".join("blk %d: LN3 bwd %d, dy2 %d, fc2 bwd products %d, L^T %d, LN2 bwd + reload %d, fc1 + shortcut bwd %d" %
          ((blk,) + tuple(st(4, blk * 8 + k + 1) - st(4, blk * 8 + k) for k in range(6))) for blk in (3, 2, 1)))
