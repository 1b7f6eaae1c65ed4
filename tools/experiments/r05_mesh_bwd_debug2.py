"""Localise a fused-backward deviation: gtape (dz, dy2, dy per GCN block) against the gradients autograd holds at the same points of the unfused chain."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pdfnet_amd import functional as F
from pdfnet_amd.networks import intaghand_decoder as D
from tests.test_meshdec_gpu import _layer

level = int(sys.argv[1]) if len(sys.argv) > 1 else 0
B = 5
layer = _layer(level, seed=5, drop=0.0).train()
V, cin = (63, 126, 252)[level], (512, 256, 128)[level]
C = cin // 2
x0 = torch.randn(2, B, V, cin, generator=torch.Generator().manual_seed(level + 3)).cuda()
gy = torch.randn(2, B, V, C, generator=torch.Generator().manual_seed(level + 4)).cuda()
# unfused with retained intermediates
F.MESH_FUSED = False
x = x0.clone().requires_grad_()
h = x + layer.position_embeddings.weight
keep = []
for i, (bl, br) in enumerate(zip(layer.graph_left.GCN_blocks, layer.graph_right.GCN_blocks)):
    xin = h
    xin.retain_grad()
    y = D._lin2(bl.fc1, br.fc1, F.cheby2_pair(xin, bl.ell, br.ell)); y.retain_grad()
    hh = D._ln2(bl.norm2, br.norm2, y, F.ACT_RELU)
    y2 = D._lin2(bl.fc2, br.fc2, F.cheby2_pair(hh, bl.ell, br.ell)); y2.retain_grad()
    s = D._lin2(bl.shortcut, br.shortcut, xin); s.retain_grad()
    _, h = D._ln2(bl.norm3, br.norm3, s, F.ACT_RELU if i != 3 else F.ACT_NONE, add=y2, p=0.0, training=True)
    keep.append((xin, y, y2, s))
out = layer.attn(h)
out.backward(gy)
F.join_wgrad()
torch.cuda.synchronize()
ref = [(a.grad.clone(), b.grad.clone(), c.grad.clone(), d.grad.clone()) for a, b, c, d in keep]
# fused
F.MESH_FUSED = True
layer.zero_grad(set_to_none=True)
xf = (x0.clone() + layer.position_embeddings.weight.detach()).requires_grad_()
cap = {}
orig = F._MeshLevel.backward
def bw(ctx, dout):
    r = orig(ctx, dout)
    return r
o2, a, tape, qkv = F.mesh_level_forward(layer, xf.detach(), training=True, save=True)
L = F._L()
dx = torch.empty_like(xf)
gt = torch.zeros(L.pdf_mesh_gtape_floats(level, B), device='cuda')
M = B * V
nws = max(2 * L.pdf_wgrad_workspace_floats(M, C, 4 * C), L.pdf_wgrad_workspace_floats(2 * M, C, C))
ws = torch.empty(nws, device='cuda')
a.dout, a.dx, a.gtape, a.wg_ws, a.wg_ws_floats = gy.data_ptr(), dx.data_ptr(), gt.data_ptr(), ws.data_ptr(), nws
tensors, slots = F._mesh_param_list(layer)
gr = [torch.zeros_like(t) for t in tensors]
for g, where in zip(gr, slots):
    for grp, i, k, hnd, wb in where:
        dst = getattr(a, grp)
        if i is not None:
            dst = dst[i]
        getattr(getattr(dst, k), wb)[hnd] = g.data_ptr()
L.pdf_mesh_level_bwd(ctypes.byref(a), F.stream(), F.stream())
torch.cuda.synchronize()
RC = 2 * B * V * C
def sl(k):
    return gt[k * RC:(k + 1) * RC].view(2, B, V, C)
for i in range(4):
    dxin_ref, dy_ref, dy2_ref, dz_ref = ref[i]
    for name, got, want in (("dz", sl(3 * i), dz_ref), ("dy2", sl(3 * i + 1), dy2_ref), ("dy", sl(3 * i + 2), dy_ref)):
        e = [(float((got[hh] - want[hh]).abs().max()), float(want[hh].abs().max())) for hh in (0, 1)]
        print("block %d %-4s left err %.2e / %.2e   right err %.2e / %.2e" % (i, name, e[0][0], e[0][1], e[1][0], e[1][1]))
e = [(float((dx[hh] - ref[0][0][hh]).abs().max()), float(ref[0][0][hh].abs().max())) for hh in (0, 1)]
print("dx           left err %.2e / %.2e   right err %.2e / %.2e" % (e[0][0], e[0][1], e[1][0], e[1][1]))
# per-sample error of dx (left)
print("dx left per sample:", [float((dx[0, b] - ref[0][0][0, b]).abs().max()) for b in range(B)])
print("dz(0) left per sample:", [float((sl(0)[0, b] - ref[0][3][0, b]).abs().max()) for b in range(B)])
d = (sl(0)[0] - ref[0][3][0]).abs()
print("dz(0) left: rows with error", (d.amax(-1) > 1e-4).nonzero().flatten().tolist()[:40], "cols", (d.amax((0, 1)) > 1e-4).nonzero().flatten().tolist()[:40])
