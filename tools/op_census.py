"""Census of the aten ops one eager train step dispatches besides the HIP library, with the pdfnet_amd call site (forward
and custom-Function backward) or the autograd node (engine-run backward) that issues them: every one is a launch the hot
path could fuse away.  Usage: python tools/op_census.py [topN]"""
import os, sys, collections, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.utils._python_dispatch import TorchDispatchMode
from bench import make_opt
from pdfnet_amd.networks.intaghand_model import load_model_intag
from pdfnet_amd.synthetic import synthetic_loss_constants, synthetic_train_batch, to_device
from pdfnet_amd.trains.simplified import CtdetLoss
from pdfnet_amd.trains.base_trainer import Trainer

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NO_LAUNCH = ('view', 'reshape', 'permute', 'transpose', 'expand', 'slice', 'select', 'as_strided', 'unsqueeze', 'squeeze',
             'detach', 'alias', 'empty', 't.default', 'unbind', 'split', 'chunk', 'narrow', 'size', 'stride', 'is_', 'numel',
             'lift_fresh', 'set_', 'resize_', 'unfold', 'movedim', 'flatten', 'contiguous', 'record_stream', 'diagonal')


class Census(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.sites = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func).replace('aten.', '')
        if not any(name.startswith(p) or ('.' + p) in name for p in NO_LAUNCH):
            site = None
            for fr in reversed(traceback.extract_stack()):
                if fr.filename.startswith(ROOT + '/pdfnet_amd'):
                    site = "%s:%d %s" % (fr.filename[len(ROOT) + 1:], fr.lineno, fr.name)
                    break
            if site is None:
                node = torch._C._current_autograd_node()
                site = 'engine: ' + (node.name() if node is not None else '?')
            self.sites[(name, site)] += 1
        return func(*args, **(kwargs or {}))


top = int(sys.argv[1]) if len(sys.argv) > 1 else 80
dev = torch.device('cuda')
opt = make_opt(256)
torch.manual_seed(0)
model = load_model_intag(opt).to(dev)
consts = synthetic_loss_constants()
trainer = Trainer(opt, model, CtdetLoss(opt, consts).to(dev), lr=1e-4, use_graph=False)
batch = to_device(synthetic_train_batch(32, 256, consts=consts), dev)
for _ in range(2):
    trainer.train_step(batch)
torch.cuda.synchronize()
c = Census()
with c:
    trainer.train_step(batch)
torch.cuda.synchronize()
ops = collections.Counter()
for (n, s), v in c.sites.items():
    ops[n] += v
print("== dispatched aten ops (views excluded): %d" % sum(ops.values()))
for k, v in ops.most_common(40):
    print("%6d  %s" % (v, k))
print("== by call site")
for (n, s), v in c.sites.most_common(top):
    print("%6d  %-30s %s" % (v, n, s[:140]))
