"""cProfile of the forward+loss issue AND the backward issue (autograd multithreading off so the backward runs on the
profiled thread) of a train step -> where the Python thread spends its time.  Usage: host_profile_bwd.py [batch]"""
import os, sys, cProfile, pstats, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bench import make_opt
from pdfnet_amd.networks.intaghand_model import load_model_intag
from pdfnet_amd.synthetic import synthetic_loss_constants, synthetic_train_batch, to_device
from pdfnet_amd.trains.simplified import CtdetLoss
from pdfnet_amd.trains.base_trainer import Trainer

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device('cuda')
opt = make_opt(256)
torch.manual_seed(0)
model = load_model_intag(opt).to(dev)
consts = synthetic_loss_constants()
trainer = Trainer(opt, model, CtdetLoss(opt, consts).to(dev), lr=1e-4)
batch = to_device(synthetic_train_batch(B, 256, consts=consts), dev)
for _ in range(5):
    trainer.train_step(batch)
torch.cuda.synchronize()
torch.autograd.set_multithreading_enabled(False)
for phase in ("forward", "backward"):
    pr = cProfile.Profile()
    tot = 0.0
    for _ in range(4):
        trainer.optimizer.zero_grad()
        if phase == "forward":
            torch.cuda.synchronize()
            t0 = time.perf_counter(); pr.enable()
        out = trainer.model_with_loss(batch, 'train', 0)
        loss = out[0].mean()
        if phase == "forward":
            pr.disable(); tot += time.perf_counter() - t0
        torch.cuda.synchronize()
        if phase == "backward":
            t0 = time.perf_counter(); pr.enable()
        loss.backward()
        if phase == "backward":
            pr.disable(); tot += time.perf_counter() - t0
        torch.cuda.synchronize()
    print("==== %s issue (profiled): %.1f ms/step" % (phase, tot / 4 * 1e3))
    st = pstats.Stats(pr)
    st.sort_stats('tottime').print_stats(38)
