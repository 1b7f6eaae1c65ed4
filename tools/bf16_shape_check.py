"""bf16 GEMM kernels vs an fp32 evaluation of bf16-rounded operands on the model's own layer shapes (GPU)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as TF
from pdfnet_amd import functional as F

rb = lambda t: t.to(torch.bfloat16).float()
F.set_gemm_precision('bf16')
g = torch.Generator().manual_seed(0)
B = 2
shapes = [(2048, 8, 128, 1, 1, 0), (128, 16, 128, 3, 1, 1), (128, 32, 128, 3, 1, 1), (128, 64, 128, 3, 1, 1), (128, 64, 42, 1, 1, 0), (128, 64, 2, 1, 1, 0),
          (1024, 64, 256, 3, 1, 1), (256, 64, 256, 3, 1, 1), (256, 64, 2, 1, 1, 0), (256, 64, 122, 1, 1, 0), (64, 64, 64, 1, 1, 0), (64, 64, 64, 3, 1, 1),
          (64, 64, 256, 1, 1, 0), (256, 64, 128, 1, 1, 0), (128, 64, 128, 3, 2, 1), (256, 64, 512, 1, 2, 0), (512, 32, 256, 3, 2, 1), (1024, 16, 512, 3, 2, 1),
          (512, 8, 2048, 1, 1, 0), (256, 5, 512, 3, 1, 0), (512, 3, 1024, 3, 1, 0)]
for cin, hw, cout, k, st, pad in shapes:
    x = rb(torch.randn(B, cin, hw, hw, generator=g)).cuda()
    w = rb(torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5).cuda()
    y = F.conv2d(x.contiguous(memory_format=torch.channels_last), w.contiguous(memory_format=torch.channels_last), None, st, pad, 0)
    ref = TF.conv2d(x.double(), w.double(), None, st, pad)
    err = float((y.double() - ref).abs().max() / ref.abs().max())
    print("conv %5d@%2d -> %4d k%d s%d  rel max err %.2e %s" % (cin, hw, cout, k, st, err, "" if err < 1e-4 else "  <<<<<<"))
for cin, hw, cout, k, st, pad in [(512, 32, 256, 4, 2, 1), (1024, 16, 256, 4, 4, 0), (2048, 8, 256, 8, 8, 0)]:
    x = rb(torch.randn(B, cin, hw, hw, generator=g)).cuda()
    w = rb(torch.randn(cin, cout, k, k, generator=g) / cin ** 0.5).cuda()
    y = F.deconv2d(x.contiguous(memory_format=torch.channels_last), w.contiguous(memory_format=torch.channels_last), None, st, pad)
    ref = TF.conv_transpose2d(x.double(), w.double(), None, st, pad)
    err = float((y.double() - ref).abs().max() / ref.abs().max())
    print("deconv %5d@%2d -> %4d k%d s%d  rel max err %.2e %s" % (cin, hw, cout, k, st, err, "" if err < 1e-4 else "  <<<<<<"))
for M, N, K in [(2 * 512 * 64, 64, 16), (2 * 512 * 64, 64, 64), (2 * 512 * 64, 128, 64), (2 * 128 * 64, 128, 144), (2 * 128 * 64, 256, 128), (256, 512, 272), (256, 1024, 512),
                (2, 509, 1024), (2 * 63, 256, 512), (2 * 63, 512, 512), (2 * 252, 64, 128), (2 * 778, 3, 64), (4, 1024, 1024), (2 * 1024, 64, 64)]:
    x = rb(torch.randn(M, K, generator=g)).cuda()
    w = rb(torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    y = F.linear(x, w, None, 0)
    ref = x.double() @ w.double().t()
    err = float((y.double() - ref).abs().max() / ref.abs().max())
    print("linear M %6d N %4d K %4d  rel max err %.2e %s" % (M, N, K, err, "" if err < 1e-4 else "  <<<<<<"))
