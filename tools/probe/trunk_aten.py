"""Which aten ops does the ResNet trunk (layer1-4) dispatch in a training forward + backward, under a Trainer (flat gradients)?"""
import collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch.utils._python_dispatch import TorchDispatchMode
from bench import make_opt
from pdfnet_amd import functional as F
from pdfnet_amd.networks.intaghand_model import load_model_intag
from pdfnet_amd.synthetic import synthetic_loss_constants
from pdfnet_amd.trains.simplified import CtdetLoss
from pdfnet_amd.trains.base_trainer import Trainer

dev = torch.device('cuda')
opt = make_opt(256)
model = load_model_intag(opt).to(dev)
tr = Trainer(opt, model, CtdetLoss(opt, synthetic_loss_constants()).to(dev), lr=1e-4)
r = model.encoder.resnet
model.train()
x = torch.randn(8, 64, 64, 64, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_()
VIEW = ('view', 'reshape', 'permute', 'transpose', 'expand', 'slice', 'select', 'squeeze', 'unsqueeze', 't.default', 'alias', 'detach', 'as_strided', '_unsafe_view', 'unbind', 'split')
ops = collections.Counter()


class Census(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        n = str(func).replace('aten.', '')
        if not any(n.startswith(v) for v in VIEW):
            ops[n] += 1
        return func(*args, **(kwargs or {}))


def run():
    x4 = r.layer1(x); x3 = r.layer2(x4); x2 = r.layer3(x3); x1 = r.layer4(x2)
    outs = (x4, x3, x2, x1)
    torch.autograd.grad(outs, [x], [torch.ones_like(o) for o in outs])
    F.join_wgrad()


run()
torch.autograd.set_multithreading_enabled(False)
with Census():
    run()
torch.cuda.synchronize()
for k, v in ops.most_common():
    print("%4d  %s" % (v, k))
