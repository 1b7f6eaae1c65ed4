"""Host-side issue time of one train step vs its GPU time: is the step launch-bound?
Prints, per step, how long the Python thread needed to ISSUE the step (no synchronisation inside) and the synchronised
time of a block of steps.  If issue time ~= step time the host is the bottleneck."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import make_opt
from pdfnet_amd.networks.intaghand_model import load_model_intag
from pdfnet_amd.synthetic import synthetic_loss_constants, synthetic_train_batch, to_device
from pdfnet_amd.trains.simplified import CtdetLoss
from pdfnet_amd.trains.base_trainer import Trainer

import pdfnet_amd.functional as F
DT = sys.argv[1] if len(sys.argv) > 1 else 'f32'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
if DT == 'bf16':
    F.set_gemm_precision('bf16')
dev = torch.device('cuda')
opt = make_opt(256)
torch.manual_seed(0)
model = load_model_intag(opt).to(dev)
consts = synthetic_loss_constants()
trainer = Trainer(opt, model, CtdetLoss(opt, consts).to(dev), lr=1e-4, grad_comm_dtype=torch.bfloat16 if DT == 'bf16' else None)
batch = to_device(synthetic_train_batch(B, 256, consts=consts), dev)
for _ in range(5):
    trainer.train_step(batch)
torch.cuda.synchronize()
N = 10
t0 = time.perf_counter()
issue = []
for _ in range(N):
    a = time.perf_counter()
    trainer.train_step(batch)
    issue.append(time.perf_counter() - a)
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(DT, "B=%d" % B)
print("issue per step: " + " ".join("%.1f" % (x * 1e3) for x in issue) + " ms")
print("host issue total %.1f ms, synchronised total %.1f ms for %d steps (%.1f ms/step)" % (t_issue * 1e3, t_all * 1e3, N, t_all / N * 1e3))
# phase split of the host time of one step
m, L = trainer.model_with_loss.model, trainer.model_with_loss.loss
torch.cuda.synchronize()
a = time.perf_counter()
trainer.optimizer.zero_grad()
out = trainer.model_with_loss(batch, 'train', 0)
b = time.perf_counter()
loss = out[0].mean()
loss.backward()
c = time.perf_counter()
torch.cuda.synchronize()
d = time.perf_counter()
print("host: forward+loss issue %.1f ms, backward issue %.1f ms, then waited %.1f ms for the GPU" % ((b - a) * 1e3, (c - b) * 1e3, (d - c) * 1e3))
