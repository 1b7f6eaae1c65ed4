"""cProfile of the forward+loss issue of one train step at B=8 (host-bound) -> where the Python thread spends its time."""
import os, sys, cProfile, pstats, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bench import make_opt
from pdfnet_amd.networks.intaghand_model import load_model_intag
from pdfnet_amd.synthetic import synthetic_loss_constants, synthetic_train_batch, to_device
from pdfnet_amd.trains.simplified import CtdetLoss
from pdfnet_amd.trains.base_trainer import Trainer

dev = torch.device('cuda')
opt = make_opt(256)
torch.manual_seed(0)
model = load_model_intag(opt).to(dev)
consts = synthetic_loss_constants()
trainer = Trainer(opt, model, CtdetLoss(opt, consts).to(dev), lr=1e-4)
batch = to_device(synthetic_train_batch(8, 256, consts=consts), dev)
for _ in range(5):
    trainer.train_step(batch)
torch.cuda.synchronize()
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
for _ in range(5):
    trainer.optimizer.zero_grad()
    out = trainer.model_with_loss(batch, 'train', 0)
    loss = out[0].mean()
    torch.cuda.synchronize()
pr.disable()
print("forward+loss (profiled, synchronised each step): %.1f ms/step" % ((time.perf_counter() - t0) / 5 * 1e3))
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(45)
