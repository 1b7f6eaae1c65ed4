"""Where does the fused backward deviate?  Relative error of dx and of every parameter gradient, fused vs unfused, in module order."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pdfnet_amd import functional as F
from tests.test_meshdec_gpu import _layer, _unfused

level = int(sys.argv[1]) if len(sys.argv) > 1 else 0
drop = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
B = 5
layer = _layer(level, seed=5, drop=drop).train()
V, cin = (63, 126, 252)[level], (512, 256, 128)[level]
x0 = torch.randn(2, B, V, cin, generator=torch.Generator().manual_seed(level + 3)).cuda()
gy = torch.randn(2, B, V, cin // 2, generator=torch.Generator().manual_seed(level + 4)).cuda()
res = {}
for fused in (False, True):
    layer.zero_grad(set_to_none=True)
    x = x0.clone().requires_grad_()
    F.manual_seed(99)
    F.MESH_FUSED = fused
    out = layer(x) if fused else _unfused(layer, x)
    F.MESH_FUSED = True
    out.backward(gy)
    F.join_wgrad()
    torch.cuda.synchronize()
    res[fused] = (x.grad.clone(), {n: p.grad.clone() for n, p in layer.named_parameters() if p.grad is not None})
(dx0, g0), (dx1, g1) = res[False], res[True]
print("dx rel err %.2e  (per hand: %.2e %.2e)" % (float((dx0 - dx1).abs().max() / dx0.abs().max()), float((dx0[0] - dx1[0]).abs().max()), float((dx0[1] - dx1[1]).abs().max())))
for n in g0:
    e = float((g0[n] - g1[n]).abs().max() / (g0[n].abs().max() + 1e-30)) if n in g1 else float('nan')
    flag = "" if e < 1e-3 else "   <-----"
    print("%-60s %.2e%s" % (n, e, flag))
