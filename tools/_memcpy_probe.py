import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from bench import make_opt
from pdfnet_amd.networks.intaghand_model import load_model_intag
from pdfnet_amd.synthetic import synthetic_loss_constants, synthetic_train_batch, to_device
from pdfnet_amd.trains.simplified import CtdetLoss
from pdfnet_amd.trains.base_trainer import Trainer
dev = torch.device('cuda')
opt = make_opt(256)
model = load_model_intag(opt).to(dev)
consts = synthetic_loss_constants()
trainer = Trainer(opt, model, CtdetLoss(opt, consts).to(dev), lr=1e-4, use_graph=False)
batch = to_device(synthetic_train_batch(32, 256, consts=consts), dev)
for _ in range(2):
    trainer.train_step(batch)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    trainer.train_step(batch)
    torch.cuda.synchronize()
cnt = collections.Counter()
names = collections.Counter()
for ev in prof.events():
    n = ev.name
    if 'emcpy' in n or 'emset' in n:
        names[n] += 1
        chain = []
        p = ev.cpu_parent
        while p is not None and len(chain) < 4:
            chain.append(p.name); p = p.cpu_parent
        cnt[(n, ' < '.join(chain))] += 1
for k, v in names.most_common(10): print(v, k)
for k, v in cnt.most_common(40): print(v, k)
