"""Allocated / reserved device memory over a few hundred train steps (fp32 B=8 to keep it short): a leak shows as a ramp."""
import os, sys
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
mode = sys.argv[1] if len(sys.argv) > 1 else 'eager'
tr = Trainer(opt, model, CtdetLoss(opt, consts).to(dev), lr=1e-4, use_graph={'eager': False, 'graph': True, 'auto': 'auto'}[mode])
batch = to_device(synthetic_train_batch(8, 256, consts=consts), dev)
for i in range(241):
    tr.train_step(batch, epoch=0 if i < 120 else 25)          # (the loss schedule switches at epoch 20: a second graph key)
    if i % 40 == 0:
        torch.cuda.synchronize()
        print("step %3d  allocated %7.1f MB  reserved %7.1f MB  use_graph=%s" % (i, torch.cuda.memory_allocated() / 1e6, torch.cuda.memory_reserved() / 1e6, tr.use_graph))
