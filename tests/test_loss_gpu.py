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


def _aten_dense(mask, mask_gt, hms, hms_gt, hm, hm_gt):
    """The reference's formulas with aten ops (checker): simplified.py:368,374,376,391; losses.py:138-165."""
    import torch.nn.functional as TF
    p = torch.clamp(torch.sigmoid(hm), 1e-4, 1 - 1e-4)
    pos, neg = hm_gt.eq(1).float(), hm_gt.lt(1).float()
    pl = (torch.log(p) * (1 - p) ** 2 * pos).sum((1, 2, 3))
    nl = (torch.log(1 - p) * p ** 2 * (1 - hm_gt) ** 4 * neg).sum((1, 2, 3))
    npos = pos.sum((1, 2, 3))
    focal = -nl if float(npos.sum()) == 0 else -(pl + nl) / (npos + 1e-3)
    return TF.smooth_l1_loss(mask, mask_gt), TF.mse_loss(hms, hms_gt), focal


@pytest.mark.parametrize("positives", [True, False])
def test_dense_loss_kernels_match_aten_formulas(positives):
    from pdfnet_amd import functional as F
    g = torch.Generator().manual_seed(3)
    B, R = 3, 64
    CL = torch.channels_last
    mask = (torch.randn(B, 2, R, R, generator=g) * 1.5).cuda().contiguous(memory_format=CL)        # |d| on both sides of 1
    mask_gt = (torch.rand(B, 2, R, R, generator=g) < 0.5).float().cuda()
    hms = torch.randn(B, 42, R // 4, R // 4, generator=g).cuda().contiguous(memory_format=CL)
    hms_gt = torch.rand(B, 42, R // 4, R // 4, generator=g).cuda()
    hm = (torch.randn(B, 2, R // 4, R // 4, generator=g) * 4).cuda().contiguous(memory_format=CL)  # some logits beyond the clamp
    hm[0, 0, 0, 0] = 30.0
    hm[0, 1, 0, 1] = -30.0
    hm_gt = (torch.rand(B, 2, R // 4, R // 4, generator=g) * 0.9).cuda()
    if positives:
        hm_gt[0, 0, 3, 4] = 1.0
        hm_gt[2, 1, 5, 5] = 1.0
        hm_gt[2, 0, 0, 0] = 1.0                                                                   # sample 1 has no positive
    w = torch.tensor([0.7, -1.3, 2.0], device='cuda')
    outs = []
    for fn in (F.dense_loss, _aten_dense):
        leaves = [t.clone().requires_grad_() for t in (mask, hms, hm)]
        a, b, c = fn(leaves[0], mask_gt, leaves[1], hms_gt, leaves[2], hm_gt)
        (3.0 * a + 5.0 * b + (c * w).sum()).backward()
        outs.append(([a.detach(), b.detach(), c.detach()], [t.grad for t in leaves]))
    for x, y in zip(outs[0][0], outs[1][0]):
        assert torch.allclose(x, y, rtol=2e-5, atol=1e-6), (x, y)
    for x, y in zip(outs[0][1], outs[1][1]):
        assert x.shape == y.shape
        assert float((x - y).abs().max()) <= 2e-5 * float(y.abs().max()) + 1e-9
    # a term without an upstream gradient is skipped
    leaves = [t.clone().requires_grad_() for t in (mask, hms, hm)]
    F.dense_loss(leaves[0], mask_gt, leaves[1], hms_gt, leaves[2], hm_gt)[1].backward()
    assert leaves[0].grad is None and leaves[2].grad is None and leaves[1].grad is not None


def test_evaluation_loop_matches_the_reference_metric_formula():
    """Trainer.evaluation (counterpart of base_trainer.py:207-429): two batches through the test-mode pass; the device-side
    accumulation must equal the reference's per-batch formula (oracle/loss_cpu.evaluation_metrics) averaged over batches."""
    from oracle import loss_cpu as LC
    from pdfnet_amd.networks.intaghand_model import load_model_intag
    from pdfnet_amd.synthetic import synthetic_loss_constants, synthetic_train_batch
    from pdfnet_amd.trains.base_trainer import Trainer, mpjpe_mm
    from pdfnet_amd.trains.simplified import CtdetLoss
    R, B = 128, 3
    dev = torch.device('cuda')
    opt = make_opt(R, size_train=[R, R], down_ratio=4, center_weight=200.0, reproj_weight=1.0, bone_dir_weight=200.0)
    consts = synthetic_loss_constants()
    torch.manual_seed(5)
    m = load_model_intag(opt).to(dev)
    tr = Trainer(opt, m, CtdetLoss(opt, consts).to(dev))
    loader = [synthetic_train_batch(B, R, seed=s, consts=consts) for s in (21, 22)]
    loader[0]['meta'] = {'not': 'a tensor'}                   # the reference's loader carries one (base_trainer.py:234-236)
    got = tr.evaluation(loader)
    assert got['samples'] == 2 * B and m.training
    want = {}
    tr.model_with_loss.eval()
    with torch.no_grad():
        for b in loader:
            bd = tree_to({k: v for k, v in b.items() if torch.is_tensor(v)}, dev)
            tup = tuple(t.cpu() for t in tr.model_with_loss(bd, 'test', None))
            for k, v in LC.evaluation_metrics(tup, (b['lms_left_gt'], b['lms_right_gt'])).items():
                want[k] = want.get(k, 0.0) + v / len(loader)
            assert abs(mpjpe_mm(tup[1].cuda(), tup[3].cuda()) - float(torch.norm(tup[1] - tup[3], dim=-1).mean()) * 1000) < 1e-2
    for k, v in want.items():
        assert abs(got[k] - v) <= 1e-4 * abs(v) + 1e-6, (k, got[k], v)
    assert abs(got['mpjpe_mm'] - (want['abs_left_joints'] + want['abs_right_joints']) / 2) <= 1e-4 * got['mpjpe_mm']


@pytest.mark.parametrize("epoch", [0, 25])
def test_fused_mesh_loss_equals_the_term_by_term_path(epoch):
    """csrc/loss.hip mesh_loss_* (round 5: the twelve mesh terms and their weighted sum in two launches forward + one backward) against the
    term-by-term kernels + aten bookkeeping it replaces -- which the test above pins to the reference's CtdetLoss: every statistic, the loss,
    and the gradient of the loss with respect to every decoder output it reads (alpha = 0 and alpha = 1: the edge / 2-D joint terms switch)."""
    from pdfnet_amd import functional as F
    from pdfnet_amd.networks.intaghand_model import load_model_intag
    from pdfnet_amd.synthetic import synthetic_loss_constants, synthetic_train_batch
    from pdfnet_amd.trains.simplified import CtdetLoss
    R, B = 256, 5
    dev = torch.device('cuda')
    opt = make_opt(R, size_train=[R, R], down_ratio=4, center_weight=200.0, reproj_weight=1.0, bone_dir_weight=200.0)
    consts = synthetic_loss_constants()
    crit = CtdetLoss(opt, consts).to(dev)
    dec = load_model_intag(opt).decoder.to(dev)
    batch = synthetic_train_batch(B, R, seed=7, consts=consts)
    batch['valid'][1, 1] = 0.0
    batch['valid'][3, 0] = 0.0
    batch = tree_to(batch, dev)
    res = {}
    for fused in (False, True):
        result, params, hand, other = tree_to(synthetic_model_outputs(B, R, 11), dev)
        other['converter_left'], other['converter_right'] = dec.converter['left'], dec.converter['right']
        # the decoder hands both hands over as halves of ONE stacked tensor: do the same, and take gradients on the stacked leaves
        leaves = {}
        for name, d in (('verts3d', result['verts3d']), ('verts2d', result['verts2d']), ('hd3', hand[0]['verts3d']), ('hd2', hand[0]['verts2d']),
                        ('root', params['root'])):
            st = torch.stack((d['left'], d['right'])).detach().requires_grad_()
            d['left'], d['right'] = st[0], st[1]
            leaves[name] = st
        F.MESH_LOSS_FUSED = fused
        try:
            loss, stats, _, _ = crit(result, params, hand, other, batch, 'train', epoch)
        finally:
            F.MESH_LOSS_FUSED = True
        w = torch.linspace(0.5, 1.5, B, device=dev)
        (loss * w).sum().backward()
        torch.cuda.synchronize()
        res[fused] = (loss.detach(), {k: torch.as_tensor(v).detach().reshape(-1) for k, v in stats.items()}, {k: v.grad.clone() for k, v in leaves.items()})
    (l0, s0, g0), (l1, s1, g1) = res[False], res[True]
    assert torch.allclose(l0, l1, rtol=2e-5, atol=1e-4), (l0, l1)
    for k in s0:
        assert torch.allclose(s0[k], s1[k], rtol=2e-5, atol=1e-6), (k, s0[k], s1[k])
    for k in g0:
        top = float(g0[k].abs().max())
        err = float((g0[k] - g1[k]).abs().max())
        assert top > 0 and err <= 2e-5 * top + 1e-7, (k, err, top)
