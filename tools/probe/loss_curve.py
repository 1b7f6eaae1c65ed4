"""Loss of the first steps of the bench's training run (fixed synthetic batch, fixed seeds, ONE fresh process per call -- seeds and the dropout step
counter are process state): how fast do the x3 and the native fp32-MFMA trajectories separate, and how far apart are two runs of the SAME mode?
(Step 0 is computed on identical weights: the difference there is the arithmetic alone.)   python tools/probe/loss_curve.py [steps] [x3 mode]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from pdfnet_amd import functional as F
from pdfnet_amd.networks.intaghand_model import load_model_intag
from pdfnet_amd.trains.simplified import CtdetLoss
from pdfnet_amd.trains.base_trainer import Trainer
from pdfnet_amd.synthetic import synthetic_loss_constants, synthetic_train_batch, to_device

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 7
dev = torch.device('cuda', 0)
F.set_x3(mode)
opt = bench.make_opt(256)
torch.manual_seed(0)
F.manual_seed(1234)
model = load_model_intag(opt).to(dev)
consts = synthetic_loss_constants()
loss = CtdetLoss(opt, consts).to(dev)
trainer = Trainer(opt, model, loss, lr=1e-4, use_graph=False)
batch = to_device(synthetic_train_batch(32, 256, seed=1, consts=consts), dev)
vals = [float(trainer.train_step(batch)) for _ in range(steps)]
print("x3 mode %d: %s" % (mode, " ".join("%.4f" % v for v in vals)))
