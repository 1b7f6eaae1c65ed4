"""Phase timeline of the EAGER step from HIP events recorded on the main stream at a dozen host points (no profiler: rocprofv3
slows the host enough to make the traced step launch-bound, which hides what the untraced step waits for).
Usage: phase_events.py [batch] [bf16]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
if len(sys.argv) > 2 and sys.argv[2] == 'bf16':
    os.environ['PDFNET_GEMM'] = 'bf16'
from bench import make_opt
from pdfnet_amd import functional as F
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
if os.environ.get('PDFNET_GEMM') == 'bf16':
    F.set_gemm_precision('bf16')                               # (the host layer's switch: it sets the library's and its own)
batch = to_device(synthetic_train_batch(B, 256, consts=consts), dev)
marks = []


def mark(name):
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    marks.append((name, e, time.perf_counter()))


enc, dec, mwl = model.encoder, model.decoder, trainer.model_with_loss
trunk0 = enc.trunk
def trunk(*a, **k):
    r = trunk0(*a, **k)
    mark('fwd: trunk + PointNet++ + fusion issued')
    r['img_fmaps'][0].register_hook(lambda g: mark('bwd: mesh decoder done (grad of fused feature)'))
    return r
enc.trunk = trunk
l4 = enc.resnet.layer4
l4_fwd = l4.forward
def l4forward(*a, **k):
    r = l4_fwd(*a, **k)
    mark('fwd: ResNet layers 1-4 (x1)')
    return r
l4.forward = l4forward
fb = enc.feat_bn
fb_fwd = fb.forward
def fbforward(*a, **k):
    r = fb_fwd(*a, **k)
    mark('fwd: pyramid + feat + feat_bn (x0)')
    if r.requires_grad:
        r.register_hook(lambda g: mark('bwd: PointNet++ / centre features / hm head done (grad of x0)'))
    return r
fb.forward = fbforward
enc.on_trunk_output_grad_prev = enc.on_trunk_output_grad
def x1_grad():
    mark('bwd: everything above the ResNet output done (grad of x1)')
    if enc.on_trunk_output_grad_prev is not None:
        enc.on_trunk_output_grad_prev()
enc.on_trunk_output_grad = x1_grad
dec_fwd = dec.forward
def dforward(*a, **k):
    r = dec_fwd(*a, **k)
    mark('fwd: mesh decoder')
    return r
dec.forward = dforward
# round 6: where the loss's backward ends and the decoder tail's begins (the gradient of the level output's first consumers, the coordinate head and the
# vertex-average head, leaves them last: the hook on the tail's INPUT fires above; this one fires when the gradient of verts3d leaves the loss)
um = dec.unsample_layer
um_fwd = um.forward
def umforward(x):
    r = um_fwd(x)
    if r.requires_grad:
        r.register_hook(lambda g: mark('bwd:   loss done, first tail gradient (grad of the up-sampled vertices)'))
    return r
um.forward = umforward
# round 5: the three DualGraphLayers on their own (forward: in / out of dual_gcn; backward: gradient of its output = loss + decoder tail done,
# gradient of its input = the levels done)
dg = dec.dual_gcn
dg_fwd = dg.forward
def dgforward(x):
    mark('fwd:   decoder head (gf layers, PE)')
    if x.requires_grad:
        x.register_hook(lambda g: mark('bwd:   three mesh levels done'))
    r = dg_fwd(x)
    mark('fwd:   three mesh levels')
    if r.requires_grad:
        r.register_hook(lambda g: mark('bwd:   loss + decoder tail done (grad of the levels\' output)'))
    return r
dg.forward = dgforward
model_fwd = model.forward
def mforward(*a, **k):
    r = model_fwd(*a, **k)
    mark('fwd: dense branches joined, mid_model')
    return r
model.forward = mforward
join0 = F.join_wgrad
opt_step = trainer.optimizer.step


def step():
    marks.clear()
    mark('start')
    trainer.reducer.reset()
    trainer.optimizer.zero_grad()
    loss, stats, _, _ = mwl(batch, 'train', 0)
    loss = loss.mean()
    mark('fwd: loss')
    loss.backward()
    model.join_deferred()
    mark('bwd: main chain + forked streams done')
    join0()
    mark('bwd: weight-gradient side streams joined')
    F.step_counter(loss.device).add_(1)
    opt_step(grad_scale=1.0)
    mark('Adam')


for _ in range(6):
    trainer.train_step(batch)
res = []
torch.cuda.synchronize()
base = torch.cuda.Event(enable_timing=True)
base.record()
torch.cuda.synchronize()
base_host = time.perf_counter()          # GPU time of an event ~ base_host + base.elapsed_time(e): `lag` = how far the GPU runs behind the host
for _ in range(6):                       # no sync between steps: the host runs ahead as in the bench loop
    step()
    res.append(list(marks))
torch.cuda.synchronize()
for r in res[-3:]:
    t0 = r[0][1]
    print("---- step (ms since its start on the main stream)")
    prev = 0.0
    for name, e, th in r[1:]:
        t = t0.elapsed_time(e)
        lag = base.elapsed_time(e) - (th - base_host) * 1e3
        print("  %8.2f  (+%6.2f)  GPU behind host issue by %7.2f ms  %s" % (t, t - prev, lag, name))
        prev = t
