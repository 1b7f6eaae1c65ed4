"""r04: where does the bf16 B=32 step time go on the host?  Per-step wall time (synchronised every step) and host issue time, eager
from the start vs Trainer(use_graph='auto') (a few eager steps, a few hipGraph steps, then the faster mode)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import torch
from bench import make_opt
from pdfnet_amd import functional as F
from pdfnet_amd.networks.intaghand_model import load_model_intag
from pdfnet_amd.synthetic import synthetic_loss_constants, synthetic_train_batch, to_device
from pdfnet_amd.trains.simplified import CtdetLoss
from pdfnet_amd.trains.base_trainer import Trainer

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
mode = sys.argv[2] if len(sys.argv) > 2 else 'eager'
dev = torch.device('cuda')
opt = make_opt(256)
F.set_gemm_precision('bf16')
torch.manual_seed(0)
model = load_model_intag(opt).to(dev)
consts = synthetic_loss_constants()
tr = Trainer(opt, model, CtdetLoss(opt, consts).to(dev), lr=1e-4, grad_comm_dtype=torch.bfloat16, use_graph='auto' if mode == 'auto' else False)
batch = to_device(synthetic_train_batch(B, 256, consts=consts), dev)
rows = []
for i in range(34):
    torch.cuda.synchronize()
    a = time.perf_counter()
    tr.train_step(batch)
    b = time.perf_counter()
    torch.cuda.synchronize()
    c = time.perf_counter()
    rows.append(((b - a) * 1e3, (c - a) * 1e3, torch.cuda.memory_reserved() >> 20))
print(mode, "B=%d" % B, "use_graph now:", tr.use_graph, getattr(tr, 'auto_choice', None))
print("issue ms:", " ".join("%.0f" % r[0] for r in rows))
print("step  ms:", " ".join("%.0f" % r[1] for r in rows))
print("reserved MiB:", " ".join("%d" % r[2] for r in rows[::4]))
