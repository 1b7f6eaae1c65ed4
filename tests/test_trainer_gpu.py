"""GPU tests of the train-loop counterpart (pdfnet_amd/trains/base_trainer.py): optimizer checkpoints interchangeable with
torch.optim.Adam (the reference's optimizer, main.py:63), hipGraph replay across the loss schedule switch at epoch 20
(simplified.py:610), gradient views that survive `model.zero_grad()`."""
import numpy as np
import pytest
import torch

from tests.util import make_opt

pytestmark = pytest.mark.gpu


def _setup(R=128, B=2, seed=7, dropout=False):
    from pdfnet_amd.networks.intaghand_model import load_model_intag
    from pdfnet_amd.synthetic import synthetic_loss_constants, synthetic_train_batch, to_device
    from pdfnet_amd.trains.simplified import CtdetLoss
    dev = torch.device('cuda')
    opt = make_opt(R, size_train=[R, R], down_ratio=4, center_weight=200.0, reproj_weight=1.0, bone_dir_weight=200.0)
    consts = synthetic_loss_constants()
    batch = to_device(synthetic_train_batch(B, R, seed=3, consts=consts), dev)
    torch.manual_seed(seed)
    m = load_model_intag(opt).to(dev)
    if not dropout:
        for mod in m.modules():
            if hasattr(mod, 'p') and isinstance(getattr(mod, 'p'), float):
                mod.p = 0.0
    return opt, m, CtdetLoss(opt, consts).to(dev), batch


def test_optimizer_checkpoint_is_torch_adam_format_both_ways(tmp_path):
    """save_model / load_model(resume=True) (lib/utils/utils.py:37-119): the optimizer entry is torch.optim.Adam's
    state_dict.  (a) ours -> torch.optim.Adam over the same parameters continues identically to ours;
    (b) a torch.optim.Adam checkpoint resumes into FlatAdam."""
    from pdfnet_amd import functional as F
    from pdfnet_amd.trains.base_trainer import ModleWithLoss, Trainer
    from pdfnet_amd.utils import load_model, save_model
    opt, ma, crit, batch = _setup()
    tr = Trainer(opt, ma, crit, lr=1e-4)
    for _ in range(2):
        tr.train_step(batch, 0)
    torch.cuda.synchronize()
    p = str(tmp_path / "model_3.pth")
    save_model(p, 3, ma, tr.optimizer)
    ck = torch.load(p)
    assert set(ck['optimizer']) == {'state', 'param_groups'}
    n_par = len(list(ma.parameters()))
    assert ck['optimizer']['param_groups'][0]['params'] == list(range(n_par))
    assert all(v['exp_avg'].is_contiguous() and tuple(v['exp_avg'].shape) == tuple(q.shape)
               for v, q in zip(ck['optimizer']['state'].values(), ma.parameters()))

    # (a) the textbook loop resumes from our checkpoint
    opt2, mb, _, _ = _setup(seed=99)
    adam = torch.optim.Adam(mb.parameters(), lr=1e-4)
    mb, adam, ep = load_model(mb, p, adam, resume=True, lr=1e-4, lr_step=[], verbose=False)
    assert ep == 3
    mwl = ModleWithLoss(mb, crit).train()
    adam.zero_grad()
    loss_b = mwl(batch, 'train', 0)[0].mean()
    loss_b.backward()
    F.join_wgrad()
    adam.step()
    loss_a = tr.train_step(batch, 0)
    torch.cuda.synchronize()
    assert abs(float(loss_a) - float(loss_b.detach())) <= 1e-5 * abs(float(loss_b.detach()))
    # moments after the resumed step agree tensor by tensor; parameters agree wherever the update is well conditioned (an
    # element whose gradient is rounding noise -- e.g. a conv bias in front of a BatchNorm -- moves by +-lr at random)
    pa = dict(ma.named_parameters())
    st_b = adam.state_dict()['state']
    o = tr.optimizer
    checked = 0
    for ci, (n, q) in enumerate(mb.named_parameters()):
        if ci not in st_b or q.dim() < 2:                     # biases in front of a normalisation hold rounding noise only
            continue
        i = o.ckpt_index[ci]
        ma_m, mb_m = o._view(o.flat_m, i), st_b[ci]['exp_avg']
        scale = float(mb_m.abs().max())
        assert float((ma_m - mb_m).abs().max()) <= 1e-4 * scale + 1e-12, n
        big = mb_m.abs() > 1e-2 * scale
        if scale > 1e-7 and big.any():
            assert float((pa[n].detach() - q.detach())[big].abs().max()) <= 5e-6, n
            checked += 1
    assert checked > 200

    # (b) FlatAdam resumes from torch.optim.Adam's checkpoint
    p2 = str(tmp_path / "model_4.pth")
    save_model(p2, 4, mb, adam)
    opt3, mc, _, _ = _setup(seed=5)
    tr3 = Trainer(opt3, mc, crit, lr=3e-4)
    load_model(mc, p2, tr3.optimizer, resume=True, lr=None, verbose=False)           # lr=None keeps the checkpoint's rate
    assert abs(tr3.optimizer.lr - 1e-4) < 1e-12
    assert float(tr3.optimizer.step_t) == 3.0
    st = adam.state_dict()['state']
    for ci, q in enumerate(mc.parameters()):
        if ci in st:
            i = tr3.optimizer.ckpt_index[ci]
            assert torch.equal(tr3.optimizer._view(tr3.optimizer.flat_m, i).cpu(), st[ci]['exp_avg'].cpu()), ci
    with pytest.raises(ValueError):
        tr3.optimizer.load_state_dict({'step': 1})           # round-1 flat layout is refused, not misread


def test_graph_replay_follows_the_loss_schedule_across_epoch_20():
    """alpha = 0 if epoch < 20 else 1 (simplified.py:610) is host state baked into a captured step: the trainer keys its
    graphs on it, so graph mode and eager mode agree on both sides of the switch.  Weights are frozen (lr = 0) so that the
    four steps of the two modes see the same function (train-mode BatchNorm uses batch statistics)."""
    from pdfnet_amd import functional as F
    from pdfnet_amd.trains.base_trainer import Trainer
    res, state = {}, {}
    for mode in ('eager', 'graph'):
        opt, m, crit, batch = _setup()
        ctr0 = int(F.step_counter(torch.device('cuda', 0)))
        tr = Trainer(opt, m, crit, lr=0.0, use_graph=(mode == 'graph'))
        old, F.ASYNC_WGRAD = F.ASYNC_WGRAD, (mode != 'graph')
        try:
            out = []
            for ep in (19, 19, 20, 20):
                loss = tr.train_step(batch, ep)
                torch.cuda.synchronize()
                out.append((float(loss), tr.optimizer.flat_g[:tr.n_live].clone()))
        finally:
            F.ASYNC_WGRAD = old
        res[mode] = out
        # state the reference advances once per batch: BatchNorm running statistics, their counters, the dropout step counter.
        # The un-captured warm-up passes of a new graph key must not advance them.
        from pdfnet_amd.networks.layers import BatchNorm
        BatchNorm.flush_counters()
        state[mode] = (m.encoder.feat_bn.running_mean.clone(), m.encoder.resnet.bn1.running_var.clone(),
                       int(m.encoder.feat_bn.num_batches_tracked), int(m.mid_model.convs[2][2].num_batches_tracked),
                       int(F.step_counter(torch.device('cuda', 0))) - ctr0)
        if mode == 'graph':
            assert len(tr._graphs) == 2
    for i in range(4):
        (le, ge), (lg, gg) = res['eager'][i], res['graph'][i]
        assert abs(le - lg) <= 1e-5 * abs(le), (i, le, lg)
        assert float((ge - gg).abs().max()) <= 1e-4 * float(ge.abs().max()), i
    assert state['eager'][2:] == state['graph'][2:] == (4, 4, 4), (state['eager'][2:], state['graph'][2:])
    for a, b in zip(state['eager'][:2], state['graph'][:2]):
        assert float((a - b).abs().max()) <= 1e-5 * float(a.abs().max()) + 1e-7
    for mode in ('eager', 'graph'):
        r = res[mode]
        assert r[2][0] > r[1][0] * 1.0001, mode                                   # alpha = 1 adds 2000*edge + 1000*joints2d
        assert float((r[2][1] - r[1][1]).abs().max()) > 1e-3 * float(r[1][1].abs().max()), mode   # and their gradients
        assert torch.equal(r[0][1], r[1][1]) or float((r[0][1] - r[1][1]).abs().max()) <= 1e-5 * float(r[0][1].abs().max())


def test_gradient_views_survive_module_zero_grad():
    """`model.zero_grad()` sets p.grad = None; the trainer re-attaches its flat views before the next backward, so the
    HIP kernels keep accumulating where Adam reads (silent stall otherwise)."""
    from pdfnet_amd.trains.base_trainer import Trainer
    opt, m, crit, batch = _setup()
    tr = Trainer(opt, m, crit, lr=1e-4)
    tr.train_step(batch, 0)
    m.zero_grad()                                            # set_to_none=True by default
    assert all(p.grad is None for p in m.parameters())
    before = tr.optimizer.flat_p.clone()
    tr.train_step(batch, 0)
    torch.cuda.synchronize()
    assert all(p.grad is not None and p.grad.data_ptr() == v.data_ptr() for p, v in zip(tr.optimizer.params, tr.optimizer._grad_views))
    assert float(tr.optimizer.flat_g[:tr.n_live].abs().sum()) > 0
    assert float((tr.optimizer.flat_p - before).abs().max()) > 1e-5
    # the never-used tail is neither reduced nor stepped
    assert float(tr.optimizer.flat_g[tr.n_live:].abs().max()) == 0.0
    assert torch.equal(tr.optimizer.flat_p[tr.n_live:], before[tr.n_live:])


def test_headline_batch_32_properties():
    """BASELINE config 3 at its real size (B = 32, 256x256; what bench.py times): size-independent properties -- every output
    finite and of the reference's shape, predicted centre indices inside the R/4 grid, the loss finite and falling over 3
    steps on a fixed batch, gradients present for every live parameter and absent for the never-used tail."""
    from pdfnet_amd.trains.base_trainer import Trainer
    R, B = 256, 32
    opt, m, crit, batch = _setup(R=R, B=B, dropout=True)
    m.eval()
    with torch.no_grad():
        result, params, hand, other = m(batch['input'], batch['choose'], batch['cloud'], batch['depth'], None, batch['K_new'], batch['valid'])
    for h in ('left', 'right'):
        assert result['verts3d'][h].shape == (B, 778, 3) and result['verts2d'][h].shape == (B, 778, 2)
        assert hand[0]['verts3d'][h].shape == (B, 252, 3)
        assert params['scale'][h].shape == (B,) and params['trans2d'][h].shape == (B, 2) and params['root'][h].shape == (B, 3)
        for t in (result['verts3d'][h], result['verts2d'][h], hand[0]['verts3d'][h], params['root'][h]):
            assert torch.isfinite(t).all()
    assert other['hms'].shape == (B, 42, R // 4, R // 4) and other['mask'].shape == (B, 2, R, R)
    assert other['ret']['hm'].shape == (B, 2, R // 4, R // 4) and other['ret']['params'].shape == (B, 122, R // 4, R // 4)
    ind = other['ind']
    assert ind.dtype == torch.int64 and ind.shape == (B, 2) and int(ind.min()) >= 0 and int(ind.max()) < (R // 4) ** 2
    tr = Trainer(opt, m, crit, lr=1e-4)
    losses = [float(tr.train_step(batch, 0)) for _ in range(3)]
    assert all(np.isfinite(losses)) and losses[2] < losses[0], losses
    assert tr.resolved_modes['batch'] == B and tr.resolved_modes['gemm'] == 'fp32' and tr.resolved_modes['launch'] == 'eager'
    g = tr.optimizer.flat_g
    assert torch.isfinite(g).all() and float(g[tr.n_live:].abs().max()) == 0.0
    live_zero = [n for (n, p) in m.named_parameters() if p.grad is not None and float(p.grad.abs().max()) == 0.0]
    from pdfnet_amd.trains.base_trainer import DEAD_PATTERN
    # besides the structurally dead tensors only the wh / params heads (no loss term, simplified.py:397-399) stay at zero
    assert all(DEAD_PATTERN.match(n) or n.startswith(('encoder.wh.', 'encoder.params.')) for n in live_zero), \
        [n for n in live_zero if not DEAD_PATTERN.match(n) and not n.startswith(('encoder.wh.', 'encoder.params.'))][:5]


def test_evaluation_writes_the_reference_score_and_submission_files(tmp_path):
    """Trainer.evaluation(score_path=, json_path=): the `H2O-val.txt` block and `hand_poses.json` of base_trainer.py:420-429,
    :328-335,486-489 -- the json holds, per action id and frame, the 2 x 21 x 3 absolute joints the test-mode pass predicted."""
    import json
    from pdfnet_amd.trains.base_trainer import Trainer
    opt, m, crit, batch = _setup(B=4)
    tr = Trainer(opt, m, crit, lr=1e-4)
    b1 = dict(batch)
    b1['id'] = torch.tensor([1, 1, 2, 2])
    b1['frame_num'] = torch.tensor([0, 1, 0, 5])
    sp, jp = str(tmp_path / "H2O-val.txt"), str(tmp_path / "hand_poses.json")
    ev = tr.evaluation([b1], score_path=sp, json_path=jp)
    lines = open(sp).read().splitlines()
    assert lines[0] == 'eval ' and lines[1] == 'abs_left_joints_loss_all: %.2f' % ev['abs_left_joints'] and len(lines) == 9
    d = json.load(open(jp))
    assert list(d) == ['modality', '1', '2'] and list(d['2']) == ['000000.txt', '000005.txt']
    m.eval()
    with torch.no_grad():
        tup = tr.model_with_loss(b1, 'test', None)
    want = tup[1][3].reshape(-1).cpu().tolist()                       # joints_pred of sample 3 = action 2, frame 5
    got = d['2']['000005.txt']
    assert len(got) == 126 and max(abs(a - b) for a, b in zip(got, want)) <= 1e-6
    with pytest.raises(KeyError):
        tr.evaluation([batch], json_path=jp)                          # no id / frame_num in the batch


def test_use_graph_auto_times_both_modes_and_keeps_one():
    """Trainer(use_graph='auto'): AUTO_SKIP + AUTO_STEPS + 1 eager steps, as many hipGraph steps, then one mode stays; the losses keep falling on
    a fixed batch through the switch (the graph replays the same step)."""
    from pdfnet_amd.networks.intaghand_model import load_model_intag
    from pdfnet_amd.synthetic import synthetic_loss_constants, synthetic_train_batch, to_device
    from pdfnet_amd.trains.simplified import CtdetLoss
    from pdfnet_amd.trains.base_trainer import Trainer
    opt = make_opt(256, size_train=[256, 256], down_ratio=4, center_weight=200.0, reproj_weight=1.0, bone_dir_weight=200.0)
    torch.manual_seed(0)
    dev = torch.device('cuda', 0)
    model = load_model_intag(opt).to(dev)
    consts = synthetic_loss_constants()
    tr = Trainer(opt, model, CtdetLoss(opt, consts).to(dev), lr=1e-4, use_graph='auto')
    batch = to_device(synthetic_train_batch(2, 256, seed=3, consts=consts), dev)
    n = 2 * (Trainer.AUTO_SKIP + Trainer.AUTO_STEPS + 1)
    losses = [float(tr.train_step(batch)) for _ in range(n + 3)]
    assert tr._auto is None and tr.use_graph in (True, False)
    assert set(tr.auto_choice) == {'eager_ms', 'graph_ms'} and all(v > 0 for v in tr.auto_choice.values())
    assert all(l == l for l in losses) and losses[-1] < losses[0]
    if not tr.use_graph:
        assert not tr._graphs


def test_taped_trunk_equals_the_eager_trunk_bit_for_bit(monkeypatch):
    """Round 5: the ResNet trunk replayed from a tape of its library calls (pdfnet_amd/taped.py) issues the same kernels with the same
    arguments on the same streams as the eager trunk.  In isolation (one external gradient per output, so no fan-in order can differ) the
    outputs, the input gradient and EVERY weight gradient in the trainer's flat buffer must equal the eager trunk's bit for bit -- on
    the first replay (right after the recording pass, whose own effects on gradients and running statistics must have been undone) and on
    the second."""
    from pdfnet_amd import functional as F
    from pdfnet_amd import taped
    from pdfnet_amd.trains.base_trainer import Trainer
    opt, m, crit, _ = _setup(R=128, B=2)
    tr = Trainer(opt, m, crit, lr=1e-4)
    m.train()
    enc = m.encoder
    bn = enc.resnet.layer3[2].bn2
    x = torch.randn(4, 64, 32, 32, device='cuda').contiguous(memory_format=torch.channels_last)
    gs, ref = None, None
    for tape in (False, True, True):
        monkeypatch.setattr(taped, 'TRUNK_TAPE', tape)
        if not tape or not enc.__dict__['_trunk_seg'].enabled:
            enc.__dict__.pop('_trunk_seg', None)               # (the segment reads the switch when it is created)
        tr.optimizer.zero_grad()
        rm0 = bn.running_mean.clone()
        xr = x.clone().requires_grad_()
        outs = enc.trunk_layers(xr)
        if gs is None:
            torch.manual_seed(1)
            gs = [torch.randn_like(o) for o in outs]
        torch.autograd.backward(outs, gs)
        F.join_wgrad()
        torch.cuda.synchronize()
        cur = ([o.detach().clone() for o in outs], xr.grad.clone(), tr.optimizer.flat_g.clone(), (bn.running_mean - rm0).clone(), rm0)
        if not tape:
            ref = cur
            continue
        seg = enc.__dict__['_trunk_seg']
        assert len(seg.entries) == 1 and all(e is not False for e in seg.entries.values())      # it WAS taped
        assert all(torch.equal(a, b) for a, b in zip(ref[0], cur[0]))
        assert torch.equal(ref[1], cur[1]) and torch.equal(ref[2], cur[2])
        assert float(cur[2].abs().max()) > 0
        # running statistics advanced exactly once per call, as in the eager trunk (the recording pass's own update was rolled back):
        # delta = momentum * (batch mean - running mean before), the batch mean taken from the eager call
        mean = ref[4] + ref[3] / bn.momentum
        assert torch.allclose(cur[3], bn.momentum * (mean - cur[4]), rtol=1e-4, atol=1e-6)


def test_taped_trunk_inside_the_train_step(monkeypatch):
    """... and inside the train step, at frozen weights (lr = 0: the optimizer must not amplify last-bit noise -- two EAGER runs of this step
    differ in the last bits of some gradients too, a few kernels add with float atomics): the loss of every step is bit-identical to the eager
    run's, the gradient after the last backward agrees to 1e-5 of its largest element, the running statistics and `num_batches_tracked`
    advance once per step, an eval pass in between leaves the tape alone."""
    from pdfnet_amd import functional as F
    from pdfnet_amd import taped
    from pdfnet_amd.networks.layers import BatchNorm
    from pdfnet_amd.trains.base_trainer import Trainer
    runs = {}
    for tape in (False, True):
        monkeypatch.setattr(taped, 'TRUNK_TAPE', tape)
        opt, m, crit, batch = _setup(R=128, B=4)
        F.manual_seed(99)
        tr = Trainer(opt, m, crit, lr=0.0)
        losses = [float(tr.train_step(batch, 0)) for _ in range(2)]
        m.eval()
        with torch.no_grad():
            m(batch['input'], batch['choose'], batch['cloud'], batch['depth'], None, batch['K_new'], batch['valid'])
        losses += [float(tr.train_step(batch, 0)) for _ in range(2)]
        torch.cuda.synchronize()
        seg = m.encoder.__dict__.get('_trunk_seg')
        assert seg is not None and (len(seg.entries) == 1 and all(e is not False for e in seg.entries.values())) == tape
        BatchNorm.flush_counters()
        bn = m.encoder.resnet.layer3[2].bn2
        runs[tape] = (losses, tr.optimizer.flat_g.clone(), bn.running_mean.clone(), bn.running_var.clone(), int(bn.num_batches_tracked))
        del tr, m
    a, b = runs[False], runs[True]
    assert a[0] == b[0] and len(set(a[0])) == 1, (a[0], b[0])
    assert float((a[1] - b[1]).abs().max()) <= 1e-5 * float(a[1].abs().max())
    assert torch.allclose(a[2], b[2], rtol=1e-5, atol=1e-7) and torch.allclose(a[3], b[3], rtol=1e-5, atol=1e-7)
    assert a[4] == b[4] == 4


def test_taped_trunk_is_not_recorded_over_accumulated_gradients(monkeypatch):
    """Round 6 guard rail: the recording pass of pdfnet_amd/taped.py ends by zeroing the segment's gradients.  If the first taped call ever happens
    in the SECOND micro-batch of an accumulation loop that would silently lose the first one -- so the recording is refused while any of those
    gradients is non-zero (the call runs eagerly, the accumulated gradient survives) and happens at the next call that finds them clean."""
    from pdfnet_amd import functional as F
    from pdfnet_amd import taped
    from pdfnet_amd.trains.base_trainer import Trainer
    monkeypatch.setattr(taped, 'TRUNK_TAPE', True)
    opt, m, crit, _ = _setup(R=128, B=2)
    tr = Trainer(opt, m, crit, lr=1e-4)
    m.train()
    enc = m.encoder
    enc.__dict__.pop('_trunk_seg', None)
    x = torch.randn(4, 64, 32, 32, device='cuda').contiguous(memory_format=torch.channels_last)
    tr.optimizer.zero_grad()
    w = enc.resnet.layer2[0].conv1.weight
    w.grad.fill_(0.25)                                           # "a previous micro-batch"
    xr = x.clone().requires_grad_()
    outs = enc.trunk_layers(xr)
    seg = enc.__dict__['_trunk_seg']
    assert len(seg.entries) == 0 and seg.pinned_bytes() == 0      # refused: nothing recorded, nothing pinned
    torch.autograd.backward(outs, [torch.ones_like(o) for o in outs])
    F.join_wgrad()
    torch.cuda.synchronize()
    assert float((w.grad - 0.25).abs().max()) > 0 and float(w.grad.abs().min()) >= 0          # accumulated ON TOP of the 0.25, not from zero
    g1 = w.grad.clone()
    tr.optimizer.zero_grad()
    xr = x.clone().requires_grad_()
    outs = enc.trunk_layers(xr)                                    # clean gradients: recorded now
    assert len(seg.entries) == 1 and all(e for e in seg.entries.values()) and seg.pinned_bytes() > 0
    torch.autograd.backward(outs, [torch.ones_like(o) for o in outs])
    F.join_wgrad()
    torch.cuda.synchronize()
    assert torch.allclose(w.grad + 0.25, g1, rtol=1e-5, atol=1e-6)
