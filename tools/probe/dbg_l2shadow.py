import torch, sys
sys.path.insert(0, '/root/repo')
from pdfnet_amd import functional as F, hip
F.set_gemm_precision('bf16')
Cs = (256, 256, 256, 256)
xs = [torch.randn(2, C, 16, 16).cuda().contiguous(memory_format=torch.channels_last).requires_grad_() for C in Cs]
ws = [(torch.rand(C).cuda() + 0.5).requires_grad_() for C in Cs]
out = F.l2norm_cat(xs, ws)
s = F.shadow_of(out)
print('fwd shadow present', s is not None, 'equal', torch.equal(s, out.detach().to(torch.bfloat16)) if s is not None else None)
gy = torch.randn_like(out)
got = {}
orig = F._L2NormCat.backward
def bw(ctx, dy):
    r = orig(ctx, dy)
    got['r'] = r
    return r
F._L2NormCat.backward = staticmethod(bw)
out.backward(gy)
for i, d in enumerate(got['r'][1:5]):
    s = F.shadow_of(d)
    ref = d.detach().to(torch.bfloat16)
    print(i, 'bwd shadow present', s is not None, 'equal', torch.equal(s, ref) if s is not None else None, 'max diff', float((s.float() - ref.float()).abs().max()) if s is not None else None)
