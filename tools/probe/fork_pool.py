"""Do forked side streams leak?  Size of functional.fork's stream pool and its busy set after train steps, after an evaluation pass, after more steps."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bench import make_opt
from pdfnet_amd import functional as F
from pdfnet_amd.networks.intaghand_model import load_model_intag
from pdfnet_amd.synthetic import synthetic_loss_constants, synthetic_train_batch, to_device
from pdfnet_amd.trains.simplified import CtdetLoss
from pdfnet_amd.trains.base_trainer import Trainer

dev = torch.device('cuda')
opt = make_opt(256)
model = load_model_intag(opt).to(dev)
consts = synthetic_loss_constants()
tr = Trainer(opt, model, CtdetLoss(opt, consts).to(dev), lr=1e-4)
batch = to_device(synthetic_train_batch(8, 256, consts=consts), dev)


def show(tag):
    torch.cuda.synchronize()
    print("%-28s fork pool %d streams, busy %s, weight-gradient side streams %d" % (tag, len(F._side.get(('fork', 0), [])), sorted(F.fork._busy.get(0, ())), len(F._wg_streams)))


show("start")
for _ in range(3):
    tr.train_step(batch)
show("after 3 train steps")
ev = tr.evaluation([synthetic_train_batch(8, 256, seed=5, consts=consts)], dev)
show("after an evaluation pass")
for _ in range(3):
    tr.train_step(batch)
show("after 3 more train steps")
