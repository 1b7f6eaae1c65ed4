"""bf16 build of the fused mesh level vs the fp32 build, and the per-op bf16 chain vs the fp32 build: relative L2 and max-norm deviations of out, dx, weight gradients."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pdfnet_amd import functional as F
from tests.test_meshdec_gpu import _layer
B = 5
for level in (0, 1, 2):
    layer = _layer(level, seed=6, drop=0.0).train()
    V, cin = (63, 126, 252)[level], (512, 256, 128)[level]
    x0 = torch.randn(2, B, V, cin, generator=torch.Generator().manual_seed(level + 5)).cuda()
    gy = torch.randn(2, B, V, cin // 2, generator=torch.Generator().manual_seed(level + 6)).cuda()
    res = {}
    for name, bf, fused in (("fp32 fused", False, True), ("bf16 fused", True, True), ("bf16 per-op", True, False), ("fp32 per-op", False, False)):
        F.set_gemm_precision('bf16' if bf else 'fp32')
        F.MESH_FUSED = fused
        layer.zero_grad(set_to_none=True)
        x = x0.clone().requires_grad_()
        F.manual_seed(99)
        out = layer(x)
        out.backward(gy)
        F.join_wgrad()
        torch.cuda.synchronize()
        res[name] = (out.detach().clone(), x.grad.clone(), torch.cat([p.grad.flatten() for n, p in layer.named_parameters() if p.grad is not None and p.dim() > 1]))
    F.set_gemm_precision('fp32'); F.MESH_FUSED = True
    ref = res["fp32 fused"]
    a_, b_ = res["bf16 per-op"], res["bf16 fused"]
    print("level %d bf16 fused vs bf16 per-op:" % level, " ".join("%s L2 %.2e max %.2e" % (k, float((a - b).norm() / a.norm()), float((a - b).abs().max() / a.abs().max())) for k, a, b in zip(("out", "dx", "dW"), a_, b_)))
    for name in ("bf16 fused", "bf16 per-op", "fp32 per-op"):
        r = res[name]
        print("level %d %-12s" % (level, name), " ".join("%s L2 %.2e max %.2e" % (k, float((a - b).norm() / a.norm()), float((a - b).abs().max() / a.abs().max())) for k, a, b in zip(("out", "dx", "dW"), ref, r)))
