"""pdf_invert_index on the two PointNet++ levels of the B=32 step (Bc = 64 clouds): random neighbours and a degenerate batch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import torch
from pdfnet_amd import hip
L = hip.lib()
for (Bc, N, S, K) in ((64, 1024, 512, 64), (64, 512, 128, 64)):
    for name, hot in (("random", N), ("16 hot points", 16)):
        idx = torch.randint(0, hot, (Bc, S * K), dtype=torch.int32, device='cuda')
        start = torch.empty((Bc, N + 1), dtype=torch.int32, device='cuda')
        lst = torch.empty((Bc, S * K), dtype=torch.int32, device='cuda')
        f = lambda: L.pdf_invert_index(idx.data_ptr(), Bc, N, S * K, start.data_ptr(), lst.data_ptr(), None, None)
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            f()
        e1.record()
        torch.cuda.synchronize()
        ok = all(torch.equal(lst[b].cpu(), torch.sort(idx[b].cpu().long(), stable=True)[1].int()) for b in (0, Bc - 1))
        print("N=%d E=%d %-14s %7.1f us per launch  correct=%s" % (N, S * K, name, e0.elapsed_time(e1) * 1e3 / 20, ok))
