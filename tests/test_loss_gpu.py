"""GPU parity of the loss counterpart (pdfnet_amd/trains/simplified.py) against the reference's own
`CtdetLoss.forward` (lib/trains/simplified.py:364-655) run through the harness (tests/golden/loss_ctdet_B3_R256.npz)."""
import numpy as np
import pytest
import torch

from tests.util import gold, make_opt, synthetic_model_outputs, tree_to

pytestmark = pytest.mark.gpu


def test_ctdet_loss_matches_reference():
    from pdfnet_amd.networks.intaghand_model import load_model_intag
    from pdfnet_amd.synthetic import synthetic_loss_constants, synthetic_train_batch
    from pdfnet_amd.trains.simplified import CtdetLoss
    g = gold("loss_ctdet_B3_R256")
    R, B = 256, 3
    dev = torch.device('cuda')
    opt = make_opt(R, size_train=[R, R], down_ratio=4, center_weight=200.0, reproj_weight=1.0, bone_dir_weight=200.0)
    consts = synthetic_loss_constants()
    crit = CtdetLoss(opt, consts).to(dev)
    dec = load_model_intag(opt).decoder.to(dev)                 # only for its vertex converters
    batch = synthetic_train_batch(B, R, seed=5, consts=consts)
    batch['valid'][1, 1] = 0.0
    batch = tree_to(batch, dev)

    def outputs():
        result, params, hand, other = tree_to(synthetic_model_outputs(B, R, 9), dev)
        other['converter_left'], other['converter_right'] = dec.converter['left'], dec.converter['right']
        return result, params, hand, other

    for epoch in (0, 25):
        loss, stats, _, _ = crit(*outputs(), batch, 'train', epoch)
        ref = g["loss_e%d" % epoch]
        assert np.allclose(loss.cpu().numpy(), ref, rtol=2e-5, atol=1e-4), (epoch, loss.cpu().numpy(), ref)
        for k, v in g.items():
            if k.startswith("stat_e%d::" % epoch):
                got = torch.as_tensor(stats[k.split("::")[1]]).reshape(-1).cpu().numpy()
                assert np.allclose(got, v, rtol=2e-5, atol=1e-6), (epoch, k, got, v)
    tup = crit(*outputs(), batch, 'test', 0)
    assert len(tup) == 9
    for i, t in enumerate(tup):
        assert np.allclose(t.cpu().numpy(), g["test%d" % i], rtol=1e-5, atol=1e-5), i


def test_loss_gradient_flows_to_every_trained_output():
    from pdfnet_amd.networks.intaghand_model import load_model_intag
    from pdfnet_amd.synthetic import synthetic_loss_constants, synthetic_train_batch
    from pdfnet_amd.trains.simplified import CtdetLoss
    R, B = 256, 2
    dev = torch.device('cuda')
    opt = make_opt(R, size_train=[R, R], down_ratio=4)
    consts = synthetic_loss_constants()
    crit = CtdetLoss(opt, consts).to(dev)
    dec = load_model_intag(opt).decoder.to(dev)
    batch = tree_to(synthetic_train_batch(B, R, seed=6, consts=consts), dev)
    result, params, hand, other = tree_to(synthetic_model_outputs(B, R, 10), dev)
    other['converter_left'], other['converter_right'] = dec.converter['left'], dec.converter['right']
    leaves = [result['verts3d']['left'], result['verts2d']['right'], params['root']['left'], hand[0]['verts3d']['right'],
              other['hms'], other['mask'], other['ret']['hm']]
    for t in leaves:
        t.requires_grad_()
    loss, _, _, _ = crit(result, params, hand, other, batch, 'train', 25)
    loss.mean().backward()
    for t in leaves:
        assert t.grad is not None and torch.isfinite(t.grad).all() and t.grad.abs().sum() > 0
    # wh / params heads get no loss term in the reference (simplified.py:397-399,613-614)
    assert other['ret']['wh'].grad is None and other['ret']['params'].grad is None
