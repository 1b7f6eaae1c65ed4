"""Does queue priority shorten the step?  The main chain (forward, backward-data, BatchNorm) bounds the fp32 step while the weight
gradients on the side streams hide under it; with the step issued on a HIGH-priority stream the side streams (created at normal
priority by functional.wgrad_stream) should yield CUs to the main chain.  python stream_priority.py [high|normal|default]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import torch
from bench import make_opt
from pdfnet_amd.networks.intaghand_model import load_model_intag
from pdfnet_amd.synthetic import synthetic_loss_constants, synthetic_train_batch, to_device
from pdfnet_amd.trains.simplified import CtdetLoss
from pdfnet_amd.trains.base_trainer import Trainer

mode = sys.argv[1] if len(sys.argv) > 1 else 'high'
dev = torch.device('cuda')
opt = make_opt(256)
torch.manual_seed(0)
model = load_model_intag(opt).to(dev)
consts = synthetic_loss_constants()
trainer = Trainer(opt, model, CtdetLoss(opt, consts).to(dev), lr=1e-4)
batch = to_device(synthetic_train_batch(32, 256, consts=consts), dev)
lo, hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, 'priority_range') else (0, -1)
s = torch.cuda.Stream(priority=-1) if mode == 'high' else torch.cuda.Stream(priority=0) if mode == 'normal' else torch.cuda.current_stream()
torch.cuda.synchronize()
with torch.cuda.stream(s):
    for _ in range(6):
        trainer.train_step(batch)
    torch.cuda.synchronize()
    N = 20
    t0 = time.perf_counter()
    for _ in range(N):
        trainer.train_step(batch)
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / N
print("main stream priority %-6s (range %s..%s): %.2f ms/step, %.1f img/s" % (mode, lo, hi, t * 1e3, 32 / t))
