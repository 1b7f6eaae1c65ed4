"""GPU: the FUSED mesh-decoder kernels (csrc/meshdec.hip -- the path that ships) against fixtures generated from the reference's own
DualGraphLayer (lib/models/networks/model_attn/DualGraph.py:21-92; gcn.py:34-110, inter_attn.py:38-125) at the real dimensions of the three
levels: V = 63 / 126 / 252, C = 512 -> 256 / 256 -> 128 / 128 -> 64, B = 3, dropout 0 (tests/golden/op_dualgraph_layer_L{0,1,2}.npz, written by
oracle/make_goldens.py op_dualgraph_layers; weights and inputs are regenerated here from the same seeded numpy generators, oracle/synth.py).

Forward (eval entry point and the autograd Function in train mode), the input gradient and every parameter gradient are compared with the
reference's values -- not with the per-op HIP chain (tests/test_meshdec_gpu.py keeps that comparison as a regression net)."""
import numpy as np
import pytest
import torch

from tests.util import gold, T

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["x3", "native"])
def _mesh_arithmetic(request):
    """Every test of this file runs twice: with the fused levels' linear products as x3 arithmetic (the shipped default, csrc/meshdec_x3.hip) and on the
    native fp32 MFMA (x3 mode bit 2 off)."""
    from pdfnet_amd import functional as F
    F.set_x3(7 if request.param == "x3" else 3)
    assert F.mesh_x3() == (request.param == "x3")
    yield
    F.set_x3(None)


def _layer(level):
    from oracle import synth
    from pdfnet_amd.networks import intaghand_decoder as D
    V, cin, cout = synth.DUALGRAPH_DIMS[level]
    gc = D.load_graph_constants()
    layer = D.DualGraphLayer(V, cin, cout, gc['ell_left'][level], gc['ell_right'][level], 4, [12, 24, 48][level], 256, (256, 128, 64)[level], 4, 0.0)
    layer.load_state_dict(synth.det_state_dict(layer.state_dict(), salt=level + 1))       # the generator's keys = the reference layer's keys (227 tensors)
    x, gy = synth.dualgraph_case(level)
    return layer.cuda(), T(x).cuda(), T(gy).cuda()


@pytest.mark.parametrize("level", [0, 1, 2])
def test_fused_level_eval_forward_equals_the_reference_layer(level):
    from pdfnet_amd import functional as F
    g = gold("op_dualgraph_layer_L%d" % level)
    layer, x, _ = _layer(level)
    layer.eval()
    with torch.no_grad():
        xin = x + layer.position_embeddings.weight
        assert F.mesh_level_ok(layer, xin), "the fused kernels must take the reference's own dimensions"
        out, _, _, _ = F.mesh_level_forward(layer, xin, training=False)
    torch.cuda.synchronize()
    ref = g["out"]
    err = float(np.abs(out.cpu().numpy() - ref).max())
    print("level %d eval: max |fused - reference| = %.2e (max |ref| %.2f)" % (level, err, float(np.abs(ref).max())))
    assert err <= 2e-5 * max(1.0, float(np.abs(ref).max()))


@pytest.mark.parametrize("level", [0, 1, 2])
def test_fused_level_train_forward_and_gradients_equal_the_reference_layer(level):
    from pdfnet_amd import functional as F
    g = gold("op_dualgraph_layer_L%d" % level)
    layer, x, gy = _layer(level)
    layer.train()
    x.requires_grad_()
    assert F.MESH_FUSED and F.mesh_level_ok(layer, x + layer.position_embeddings.weight)
    out = layer(x)
    out.backward(gy)
    F.join_wgrad()
    torch.cuda.synchronize()
    ref = g["out"]
    err = float(np.abs(out.detach().cpu().numpy() - ref).max())
    assert err <= 2e-5 * max(1.0, float(np.abs(ref).max())), err
    # input gradient: every level::8-th feature element for element, plus the norm of each (hand, sample) slice
    dx = x.grad.cpu()
    sub, top = dx[..., level::8].numpy(), float(np.abs(g["dx_sub"]).max())
    e_dx = float(np.abs(sub - g["dx_sub"]).max())
    nrm = dx.double().flatten(2).norm(dim=2).numpy()
    e_n = float(np.abs(nrm / g["dx_norm"] - 1).max())
    print("level %d train: out %.2e; dx max |diff| %.2e of max %.2e; slice norms within %.1e" % (level, err, e_dx, top, e_n))
    assert e_dx <= 2e-4 * top and e_n <= 1e-4
    # parameter gradients: the set the reference produces (norm1 and img_ex get none), norm to 1e-3 relative (App. C) and
    # the first 64 elements of each against the tensor's own largest of them
    mine = {n: p.grad for n, p in layer.named_parameters() if p.grad is not None}
    names = [str(n) for n in g["grad_names"]]
    assert set(mine) == set(names), set(mine) ^ set(names)
    worst = []
    for i, n in enumerate(names):
        gr = mine[n].detach().cpu()
        head = gr.flatten()[:64].numpy()
        rh = g["grad_head"][i][:head.size]
        rn = float(g["grad_norm"][i])
        if n.endswith("w_ks.bias"):                          # zero in exact arithmetic (softmax is shift-invariant): rounding noise on both sides
            assert float(gr.abs().max()) <= 1e-4 * float(mine[n[:-4] + "weight"].abs().max()) + 1e-7
            continue
        e_norm = abs(float(gr.double().norm()) / rn - 1)
        e_head = float(np.abs(head - rh).max()) / (float(np.abs(rh).max()) + 1e-30)
        worst.append((max(e_norm, e_head), n))
        assert e_norm <= 1e-3 and e_head <= 2e-3, (n, e_norm, e_head)
    worst.sort(reverse=True)
    print("level %d: %d parameter gradients against the reference; worst %s" % (level, len(names), ["%s %.1e" % (n, e) for e, n in worst[:3]]))
