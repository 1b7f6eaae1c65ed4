"""The taped trunk against the eager trunk in isolation: outputs, input gradient and every weight gradient (one external gradient per output: the
fan-in additions have two addends, so the order cannot matter and everything must be bit-identical)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bench import make_opt
from pdfnet_amd import functional as F, taped
from pdfnet_amd.networks.intaghand_model import load_model_intag
from pdfnet_amd.synthetic import synthetic_loss_constants
from pdfnet_amd.trains.simplified import CtdetLoss
from pdfnet_amd.trains.base_trainer import Trainer
dev = torch.device('cuda')
opt = make_opt(128)
torch.manual_seed(0)
model = load_model_intag(opt).to(dev)
tr = Trainer(opt, model, CtdetLoss(opt, synthetic_loss_constants()).to(dev), lr=1e-4)
model.train()
enc = model.encoder
x = torch.randn(4, 64, 32, 32, device=dev).contiguous(memory_format=torch.channels_last)
gs = None
res = {}
for tape in (False, True, True):
    taped.TRUNK_TAPE = tape
    if not tape:
        enc.__dict__.pop('_trunk_seg', None)
    elif '_trunk_seg' in enc.__dict__ and not enc.__dict__['_trunk_seg'].enabled:
        del enc.__dict__['_trunk_seg']
    tr.optimizer.zero_grad()
    xr = x.clone().requires_grad_()
    outs = enc.trunk_layers(xr)
    if gs is None:
        torch.manual_seed(1)
        gs = [torch.randn_like(o) for o in outs]
    torch.autograd.backward(outs, gs)
    F.join_wgrad()
    torch.cuda.synchronize()
    cur = ([o.detach().clone() for o in outs], xr.grad.clone(), tr.optimizer.flat_g.clone())
    if not tape:
        ref = cur
    else:
        print("taped: outs equal", all(torch.equal(a, b) for a, b in zip(ref[0], cur[0])), "| dx equal", torch.equal(ref[1], cur[1]),
              "max diff %.3g" % float((ref[1] - cur[1]).abs().max()), "| flat gradient equal", torch.equal(ref[2], cur[2]),
              "max diff %.3g of %.3g" % (float((ref[2] - cur[2]).abs().max()), float(ref[2].abs().max())))
