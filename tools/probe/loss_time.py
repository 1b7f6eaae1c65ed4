"""GPU time of CtdetLoss forward + backward alone (the model's outputs detached): how much of the step's dependent chain is the
loss and its glue.  Usage: loss_time.py [batch]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bench import make_opt
from pdfnet_amd import functional as F
from pdfnet_amd.networks.intaghand_model import load_model_intag
from pdfnet_amd.synthetic import synthetic_loss_constants, synthetic_train_batch, to_device
from pdfnet_amd.trains.simplified import CtdetLoss

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device('cuda')
opt = make_opt(256)
torch.manual_seed(0)
model = load_model_intag(opt).to(dev).train()
consts = synthetic_loss_constants()
lossm = CtdetLoss(opt, consts).to(dev)
batch = to_device(synthetic_train_batch(B, 256, consts=consts), dev)
with torch.no_grad():
    out = model(batch['input'], batch['choose'], batch['cloud'], batch['depth'], batch['ind'], batch['K_new'], batch['valid'])


def detach(o):
    if torch.is_tensor(o):
        return o.detach().clone().requires_grad_(o.is_floating_point())
    if isinstance(o, dict):
        return {k: detach(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return type(o)(detach(v) for v in o)
    return o


def run():
    o = detach(out)
    torch.cuda.synchronize()
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    e0.record()
    loss, _, _, _ = lossm(*o, batch, 'train', 25)
    l = loss.mean()
    e1.record()
    l.backward()
    F.join_wgrad()
    e2.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1), e1.elapsed_time(e2)


for _ in range(3):
    run()
f, b = zip(*[run() for _ in range(10)])
print("CtdetLoss alone, B=%d: forward %.3f ms, backward %.3f ms (medians of 10; the host is not ahead here: upper bounds)" % (B, sorted(f)[5], sorted(b)[5]))
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    run()
ev = [e for e in prof.events() if e.device_type is not None and 'cuda' in str(e.device_type).lower()]
tot = sum(e.device_time for e in ev)
print("kernels in one forward+backward: %d, summed kernel time %.3f ms" % (len(ev), tot / 1e3))
import collections, re
cnt = collections.Counter(re.sub(r"<.*|\(.*", "", e.name)[:60] for e in ev)
for k, v in cnt.most_common(25):
    print("%4d  %s" % (v, k))
