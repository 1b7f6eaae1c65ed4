import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests.util import gold, make_opt, pack_outputs, surrogate_loss
from oracle import synth
from pdfnet_amd.networks.intaghand_model import load_model_intag
m = load_model_intag(make_opt(256))
sd = synth.det_state_dict(m.state_dict()); m.load_state_dict(sd); m.cuda()
for mod in m.modules():
    if isinstance(getattr(mod, 'p', None), float): mod.p = 0.0
b = synth.to_torch(synth.synthetic_batch(2, 256, seed=1, variant='mixed'), 'cuda')
g = gold("e2e_train_B2_R256")
m.train(); m.zero_grad()
res = m(b['input'], b['choose'], b['cloud'], b['depth'], b['ind'], b['K_new'], b['valid'])
loss = surrogate_loss(res); print('loss', loss.item(), float(g['loss'][0]))
loss.backward()
named = dict(m.named_parameters())
for k, v in g.items():
    if k.startswith("gradnorm::"):
        name = k[10:]; gr = named[name].grad
        n = gr.double().norm().item()
        head = gr.contiguous().flatten()[:64].cpu().numpy()
        ref = g["gradhead::" + name]
        print("%-75s norm rel %.2e  head rel %.2e  (|ref| %.2e)" % (name, abs(n - float(v[0])) / float(v[0]), np.abs(head - ref).max() / (np.abs(ref).max() + 1e-30), np.abs(ref).max()))
print('nograd', sum(p.grad is None for p in named.values()), int(g["n_params_without_grad"][0]))
