"""Eager-mode wall time of the train step's segments (events on the current stream)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import make_opt
from pdfnet_amd import functional as F
from pdfnet_amd.networks.intaghand_model import load_model_intag
from pdfnet_amd.synthetic import synthetic_loss_constants, synthetic_train_batch, to_device
from pdfnet_amd.trains.simplified import CtdetLoss
from pdfnet_amd.trains.base_trainer import FlatAdam

dev = torch.device('cuda')
opt = make_opt(256)
torch.manual_seed(0)
m = load_model_intag(opt).to(dev)
consts = synthetic_loss_constants()
lossm = CtdetLoss(opt, consts).to(dev)
adam = FlatAdam(m.parameters())
b = to_device(synthetic_train_batch(32, 256, consts=consts), dev)
m.train()
marks = []
def mark(name):
    e = torch.cuda.Event(enable_timing=True); e.record(); marks.append((name, e))

enc, dec = m.encoder, m.decoder
import types
def run():
    marks.clear()
    adam.zero_grad()
    mark('start')
    r = enc.resnet
    img = F.cl(b['input'])
    emb0 = enc.e_conv1(img, F.ACT_RELU)
    emb1 = r.bn1(r.conv1(img), relu=True)
    x4 = r.layer1(F.maxpool3s2(emb1)); x3 = r.layer2(x4); x2 = r.layer3(x3); x1 = r.layer4(x2)
    mark('resnet')
    pyr = torch.cat([enc.p2_l2(enc.p2(x4)), enc.p3_l2(enc.p3(x3)), enc.p4_l2(enc.p4(x2)), enc.p5_l2(enc.p5(x1))], 1)
    x0 = enc.feat_bn(enc.feat(pyr), relu=True)
    mark('pyramid+feat')
    ret = {}
    for head in opt.heads:
        fc = getattr(enc, head); ret[head] = fc[2](fc[0](x0, F.ACT_RELU))
    mark('heads')
    (hms, hms_f), (mask, dp_f) = F.parallel(lambda: enc.hms_decoder(x1), lambda: enc.dp_decoder(x1))
    mark('hms/dp decoders')
    center = enc.center_features(x0, b['ind'])
    mark('center')
    emb = [emb0, emb1, x0]
    fl = enc.pointnet_plus(b['cloud'][:, 0], emb, b['choose'][:, 0]); fr = enc.pointnet_plus(b['cloud'][:, 1], emb, b['choose'][:, 1])
    fuse = enc.sft(torch.cat((fl, fr), 1), center)
    mark('pointnet+sft')
    gl, gr, _ = m.mid_model([fuse, x2, x3, x4], hms_f, dp_f)
    mark('mid_model')
    result, pd, hl, other = dec(gl, gr)
    mark('gcn decoder')
    other.update(hms=hms, mask=mask, ret=ret, converter_left=dec.converter['left'], converter_right=dec.converter['right'])
    loss, stats, _, _ = lossm(result, pd, hl, other, b, 'train', 0)
    loss = loss.mean()
    mark('loss fwd')
    loss.backward()
    mark('backward (all)')
    adam.step()
    mark('adam')
    torch.cuda.synchronize()

for _ in range(3): run()
tot = marks[0][1].elapsed_time(marks[-1][1])
for (n0, e0), (n1, e1) in zip(marks[:-1], marks[1:]):
    print("%-20s %8.2f ms" % (n1, e0.elapsed_time(e1)))
print("total %.2f ms" % tot)
# backward split: time backward of decoder-only by differentiating a decoder-only graph
gl = torch.randn(32, 1024, device=dev, requires_grad=True); gr = torch.randn(32, 1024, device=dev, requires_grad=True)
for _ in range(2):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e2 = torch.cuda.Event(enable_timing=True)
    e0.record(); res = dec(gl, gr); e1.record()
    (res[0]['verts3d']['left'].sum() + res[0]['verts3d']['right'].sum() + res[1]['root']['left'].sum() + res[1]['root']['right'].sum()).backward(); e2.record()
    torch.cuda.synchronize()
print("decoder alone: fwd %.2f ms, bwd %.2f ms" % (e0.elapsed_time(e1), e1.elapsed_time(e2)))
