"""Stage clocks of the fused mesh decoder's forward (diagnostic build -DMD_STAMPS=1 loaded through PDFNET_HIP_LIB): microseconds per stage of workgroup 0."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pdfnet_amd import functional as F
from pdfnet_amd.networks import intaghand_decoder as D
B = 32
g = D.load_graph_constants()
L = F._L()
buf = (ctypes.c_ulonglong * (8 * 3 * 64))()
names0 = ['setup', 'fc1+sc products (loads, Lx)', 'epilogue', 'LN2', 'L h', 'fc2 product', 'z epilogue', 'LN3']
for level in (0, 1, 2):
    V, cin, cout = (63, 126, 252)[level], (512, 256, 128)[level], (256, 128, 64)[level]
    layer = D.DualGraphLayer(V, cin, cout, g['ell_left'][level], g['ell_right'][level], 4, [12, 24, 48][level], 256, cout, 4, 0.05).cuda().eval()
    x = torch.randn(2, B, V, cin, device='cuda')
    with torch.no_grad():
        for _ in range(3):
            F.mesh_level_forward(layer, x, training=False)
    torch.cuda.synchronize()
    assert L.pdf_debug_mesh_stamps(buf)
    clk = 100e6                                         # s_memtime / readcyclecounter: 100 MHz constant clock
    def t(k, n):
        return buf[(k * 3 + level) * 64 + n]
    print("level %d, gcn kernel (us since start): " % level)
    for blk in range(4):
        prev = t(0, 1 + blk * 8)
        row = []
        for n in range(2, 9):
            cur = t(0, n + blk * 8)
            row.append("%s %.1f" % (names0[n - 1], (cur - prev) / clk * 1e6))
            prev = cur
        print("   block %d: " % blk + "; ".join(row))
    print("   q/k/v: %.1f; total %.1f" % ((t(0, 40) - t(0, 33)) / clk * 1e6, (t(0, 40) - t(0, 0)) / clk * 1e6))
    print("   attention kernel (last launched = cross): K,V loads %.1f; attention %.1f; fc + z %.1f; LN %.1f; f1 f2 %.1f" % tuple((t(1, n + 1) - t(1, n)) / clk * 1e6 for n in range(5)))

# ---- backward
print("backward (k cycles of workgroup 0):")
for level in (0, 1, 2):
    V, cin, cout = (63, 126, 252)[level], (512, 256, 128)[level], (256, 128, 64)[level]
    layer = D.DualGraphLayer(V, cin, cout, g['ell_left'][level], g['ell_right'][level], 4, [12, 24, 48][level], 256, cout, 4, 0.05).cuda().train()
    x = torch.randn(2, B, V, cin, device='cuda')
    gy = torch.randn(2, B, V, cout, device='cuda')
    for _ in range(2):
        xr = x.clone().requires_grad_()
        layer(xr).backward(gy)
        F.join_wgrad()
    torch.cuda.synchronize()
    assert L.pdf_debug_mesh_stamps(buf)
    def t(k, n):
        return buf[(k * 3 + level) * 64 + n]
    def seq(k, ns, names):
        return "; ".join("%s %.1f" % (nm, (t(k, b) - t(k, a_)) / 1e3) for (a_, b), nm in zip(zip(ns[:-1], ns[1:]), names))
    print("level %d tail bwd (self): " % level + seq(2, list(range(9)), ['du', 'f2 product', 'dt_pre epi', 'f1 product', 'to LDS', 'LN bwd', 'do', 'fc product+store']) + "; total %.1f" % ((t(2, 8) - t(2, 0)) / 1e3))
    print("level %d attention bwd (self): " % level + seq(3, list(range(6)), ['loads+dq', 'loads+D', 'dk dv', '3 products', 'LN bwd']) + "; total %.1f" % ((t(3, 5) - t(3, 0)) / 1e3))
    for blk in (3, 1):
        print("level %d gcn bwd block %d: " % (level, blk) + seq(4, [blk * 8 + i for i in range(7)], ['LN3 bwd', 'dy2', 'fc2 bwd products', 'L^T', 'LN2 bwd+reload', 'fc1+sc bwd']))
    print("level %d gcn bwd block 0: " % level + seq(4, [0, 1, 2, 3, 4, 5, 6], ['LN3 bwd', 'dy2', 'fc2 bwd products', 'L^T', 'LN2 bwd+reload', 'fc1+sc bwd (2 chunks)']) + "; kernel total %.1f" % ((t(4, 6) - t(4, 24)) / 1e3))
