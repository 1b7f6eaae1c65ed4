import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import torch
from pdfnet_amd import hip
from pdfnet_amd.hip import ptr, stream
L = hip.lib()
CL = torch.channels_last
for (N, Cin, H, Cout) in ((8, 256, 64, 256), (16, 256, 64, 256), (32, 256, 64, 256), (32, 1024, 64, 256)):
    g = torch.Generator().manual_seed(1)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) * (Cin * 9) ** -0.5).cuda().contiguous(memory_format=CL)
    dy = torch.randn(N, Cout, H, H, generator=g).cuda().contiguous(memory_format=CL)
    outs = []
    for use in (0, 1, 1):
        dx = torch.full((N, Cin, H, H), 7.0, device='cuda').contiguous(memory_format=CL)
        n = L.pdf_conv2d_winograd_workspace_floats(N, H, H, Cin, Cout, 3, 3, 1, 1, 1)
        ws = torch.empty(n, device='cuda')
        o = hip.CallOpts(ws=ptr(ws) if use else None, ws_floats=n if use else 0)
        L.pdf_conv2d_bwd_data_x(ptr(dy), ptr(w), ptr(dx), N, H, H, Cin, Cin, Cout, 3, 3, 1, 1, H, H, Cout, stream(), ctypes.byref(o))
        torch.cuda.synchronize()
        outs.append(dx)
    d = (outs[1] - outs[0]).abs()
    d2 = (outs[2] - outs[1]).abs()
    bad = (d > 1e-3)
    print(N, Cin, Cout, "max diff wino-direct %.3e, wino run-to-run %.3e, bad elems %d of %d" % (float(d.max()), float(d2.max()), int(bad.sum()), d.numel()))
    if bad.any():
        idx = bad.permute(0, 2, 3, 1).nonzero()      # n, y, x, c
        print("  n range", int(idx[:, 0].min()), int(idx[:, 0].max()), " y", int(idx[:, 1].min()), int(idx[:, 1].max()), " x", int(idx[:, 2].min()), int(idx[:, 2].max()),
              " c", int(idx[:, 3].min()), int(idx[:, 3].max()), " first", idx[:5].tolist())
