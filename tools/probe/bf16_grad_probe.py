"""bf16 vs fp32 gradient agreement of the RGB-only encoder (random init, B=8): per-layer cosine."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bench import make_opt
from pdfnet_amd import functional as F
from pdfnet_amd.networks.intaghand_model import load_model_intag
torch.manual_seed(0)
enc = load_model_intag(make_opt(256)).encoder.cuda().train()
img = torch.randn(8, 3, 256, 256, device='cuda')
g = {}
for mode in ('fp32', 'bf16', 'fp32b'):
    F.set_gemm_precision('bf16' if mode == 'bf16' else 'fp32')
    enc.zero_grad(set_to_none=True)
    x = img if mode != 'fp32b' else img * (1 + 1e-3 * torch.randn_like(img))     # fp32 with a 0.1 % input perturbation: the net's own sensitivity
    x0, emb0, x1 = enc.rgb_encoder(x)
    (x0.pow(2).mean() + x1.pow(2).mean()).backward()
    F.join_wgrad(); torch.cuda.synchronize()
    g[mode] = {n: p.grad.clone() for n, p in enc.named_parameters() if p.grad is not None and p.dim() == 4}
for n in list(g['fp32'])[:6] + list(g['fp32'])[-8:]:
    a, b, c = g['fp32'][n], g['bf16'][n], g['fp32b'][n]
    cos = lambda u, v: float((u * v).sum() / (u.norm() * v.norm()))
    print("%-44s cos(bf16,fp32) %.4f   cos(fp32 perturbed 1e-3, fp32) %.4f" % (n, cos(a, b), cos(a, c)))
