"""BASELINE.json configs at their REAL per-GPU sizes -- what `bench.py` times, as `-m gpu` tests:

  configs[3]  B=256 over 8 GPUs, bf16  -> the per-rank step: B=32, bf16 GEMMs, bf16 gradient transport over RCCL
  configs[4]  B=512 over 8 GPUs, bf16  -> the per-rank step: B=64, same

(configs[2], fp32 B=32, is tests/test_trainer_gpu.py::test_headline_batch_32_properties; configs[1] at 256x256 is
tests/test_model_gpu.py::test_rgb_only_encoder_config2_forward_and_gradients[256]; configs[0] is the demo golden.)
At these sizes the CPU oracle does not finish in seconds, so the checks are the size-independent properties of the domain:
shapes of the reference's result structure, finiteness, predicted centre indices inside the R/4 grid, which gradients are
zero, the loss falling on a fixed batch -- plus, for bf16, agreement of the eval-mode meshes with the fp32 kernels of the same
library on the same weights (bf16's own error bar) and the bf16 all-reduce really going through RCCL on a
one-rank group."""
import os

import numpy as np
import pytest
import torch

from tests.util import make_opt

pytestmark = pytest.mark.gpu


def _one_rank_group():
    import torch.distributed as dist
    if dist.is_initialized():
        return False
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29541')
    dist.init_process_group('nccl', rank=0, world_size=1)
    return True


@pytest.mark.parametrize("B", [32, 64])
def test_bf16_per_rank_step_properties(B):
    """configs[3] (B=32 per GPU) and configs[4] (B=64 per GPU): bf16 MFMA GEMMs + bf16 shadows, fp32 master weights,
    gradients reduced as bf16 over a (one-rank) RCCL group, early slice from inside the backward."""
    import torch.distributed as dist
    from pdfnet_amd import functional as F
    from pdfnet_amd.networks.intaghand_model import load_model_intag
    from pdfnet_amd.synthetic import synthetic_loss_constants, synthetic_train_batch, to_device
    from pdfnet_amd.trains.base_trainer import DEAD_PATTERN, Trainer
    from pdfnet_amd.trains.simplified import CtdetLoss
    R = 256
    dev = torch.device('cuda')
    opt = make_opt(R, size_train=[R, R], down_ratio=4, center_weight=200.0, reproj_weight=1.0, bone_dir_weight=200.0)
    consts = synthetic_loss_constants()
    batch = to_device(synthetic_train_batch(B, R, seed=5, consts=consts), dev)
    torch.manual_seed(3)
    m = load_model_intag(opt).to(dev)
    created = _one_rank_group()
    try:
        # eval-mode meshes: bf16 kernels vs the fp32 kernels of the same library, same weights and inputs
        m.eval()
        outs = {}
        for mode in ('fp32', 'bf16'):
            F.set_gemm_precision(mode)
            with torch.no_grad():
                result, params, hand, other = m(batch['input'], batch['choose'], batch['cloud'], batch['depth'], None, batch['K_new'], batch['valid'])
            outs[mode] = (result, params, hand, other)
        result, params, hand, other = outs['bf16']
        for h in ('left', 'right'):
            assert result['verts3d'][h].shape == (B, 778, 3) and result['verts2d'][h].shape == (B, 778, 2)
            assert hand[0]['verts3d'][h].shape == (B, 252, 3)
            assert params['scale'][h].shape == (B,) and params['trans2d'][h].shape == (B, 2) and params['root'][h].shape == (B, 3)
            for t in (result['verts3d'][h], result['verts2d'][h], hand[0]['verts3d'][h], params['root'][h]):
                assert torch.isfinite(t).all()
        assert other['hms'].shape == (B, 42, R // 4, R // 4) and other['mask'].shape == (B, 2, R, R)
        ind = other['ind']
        assert ind.dtype == torch.int64 and ind.shape == (B, 2) and int(ind.min()) >= 0 and int(ind.max()) < (R // 4) ** 2
        # the dense maps feed no argmax: bf16 vs fp32 within bf16's error through ~60 layers (1e-2 relative)
        for k in ('hms', 'mask'):
            a, b = outs['bf16'][3][k].double(), outs['fp32'][3][k].double()
            assert float((a - b).norm() / (b.norm() + 1e-30)) <= 3e-2, k
        # samples whose predicted centres agree in both precisions see the same sparse windows: their meshes agree to bf16 accuracy
        same = (outs['bf16'][3]['ind'] == outs['fp32'][3]['ind']).all(1)
        assert int(same.sum()) >= B // 2, int(same.sum())
        for h in ('left', 'right'):
            a, b = outs['bf16'][0]['verts3d'][h][same].double(), outs['fp32'][0]['verts3d'][h][same].double()
            assert float((a - b).norm() / (b.norm() + 1e-30)) <= 5e-2, h

        F.set_gemm_precision('bf16')
        tr = Trainer(opt, m, CtdetLoss(opt, consts).to(dev), lr=1e-4, grad_comm_dtype=torch.bfloat16)
        tr.force_collectives = True                                   # the RCCL calls are issued for real on the one-rank group
        n0 = F._L().pdf_debug_shadow_operands()
        losses = [float(tr.train_step(batch, 0)) for _ in range(3)]
        torch.cuda.synchronize()
        assert tr.reducer.early is not None and len(tr.reducer.early) == 3          # early slice reduced from inside the backward
        assert tr.reducer.bytes_sent == tr.n_live * 2                                # bf16 on the wire
        assert F._L().pdf_debug_shadow_operands() > n0                               # the bf16 shadows were consumed by the GEMMs
        assert all(np.isfinite(losses)) and losses[2] < losses[0], losses
        g = tr.optimizer.flat_g
        assert torch.isfinite(g).all() and float(g[tr.n_live:].abs().max()) == 0.0
        live_zero = [n for (n, p) in m.named_parameters() if p.grad is not None and float(p.grad.abs().max()) == 0.0]
        assert all(DEAD_PATTERN.match(n) or n.startswith(('encoder.wh.', 'encoder.params.')) for n in live_zero), \
            [n for n in live_zero if not DEAD_PATTERN.match(n) and not n.startswith(('encoder.wh.', 'encoder.params.'))][:5]
        # the gradient that went through the bf16 all-reduce is the bf16 rounding of the local one (one rank: sum of one)
        gb = g[:tr.n_live]
        assert torch.equal(gb, gb.to(torch.bfloat16).float())
    finally:
        F.set_gemm_precision('fp32')
        if created:
            dist.destroy_process_group()


def test_bf16_evaluation_after_a_train_step_sees_the_updated_weights():
    """FlatAdam.step() updates the flat master buffer through raw pointers, so the bf16 weight shadows cast
    before the update must not survive it -- evaluation right after a train step has to run on the NEW weights.  Compared with
    the same evaluation with shadows off (the GEMMs then round the updated fp32 weights themselves: bit-identical by design)."""
    from pdfnet_amd import functional as F
    from pdfnet_amd.networks.intaghand_model import load_model_intag
    from pdfnet_amd.synthetic import synthetic_loss_constants, synthetic_train_batch, to_device
    from pdfnet_amd.trains.base_trainer import Trainer
    from pdfnet_amd.trains.simplified import CtdetLoss
    R, B = 128, 2
    dev = torch.device('cuda')
    opt = make_opt(R, size_train=[R, R], down_ratio=4, center_weight=200.0, reproj_weight=1.0, bone_dir_weight=200.0)
    consts = synthetic_loss_constants()
    batch = to_device(synthetic_train_batch(B, R, seed=9, consts=consts), dev)
    torch.manual_seed(5)
    m = load_model_intag(opt).to(dev)
    F.set_gemm_precision('bf16')
    try:
        tr = Trainer(opt, m, CtdetLoss(opt, consts).to(dev), lr=1e-2)        # a large step: stale weights would be far off
        tr.train_step(batch, 0)
        tr.train_step(batch, 0)
        m.eval()

        def run():
            with torch.no_grad():
                return m(batch['input'], batch['choose'], batch['cloud'], batch['depth'], batch['ind'], batch['K_new'], batch['valid'])
        with_shadows = run()
        old, F.BF16_SHADOWS = F.BF16_SHADOWS, False
        try:
            without = run()
        finally:
            F.BF16_SHADOWS = old
        for h in ('left', 'right'):
            assert torch.equal(with_shadows[0]['verts3d'][h], without[0]['verts3d'][h]), h
        assert torch.equal(with_shadows[3]['hms'], without[3]['hms'])
        ev = tr.evaluation([batch])
        assert np.isfinite(ev['mpjpe_mm'])
    finally:
        F.set_gemm_precision('fp32')


def test_precision_toggled_on_a_live_trainer_never_serves_stale_bf16_weights():
    """train in bf16, switch to fp32 (step() then updates the master buffer without re-casting the bf16 copy),
    switch back to bf16: a plain model call, an evaluation and the next train step must all see the CURRENT weights --
    `flat_p16` equals the bf16 cast of `flat_p` whenever a shadow is attached."""
    from pdfnet_amd import functional as F
    from pdfnet_amd.networks.intaghand_model import load_model_intag
    from pdfnet_amd.synthetic import synthetic_loss_constants, synthetic_train_batch, to_device
    from pdfnet_amd.trains.base_trainer import Trainer
    from pdfnet_amd.trains.simplified import CtdetLoss
    R, B = 128, 2
    dev = torch.device('cuda')
    opt = make_opt(R, size_train=[R, R], down_ratio=4, center_weight=200.0, reproj_weight=1.0, bone_dir_weight=200.0)
    consts = synthetic_loss_constants()
    batch = to_device(synthetic_train_batch(B, R, seed=9, consts=consts), dev)
    torch.manual_seed(5)
    m = load_model_intag(opt).to(dev)

    def run():
        m.eval()
        with torch.no_grad():
            return m(batch['input'], batch['choose'], batch['cloud'], batch['depth'], batch['ind'], batch['K_new'], batch['valid'])
    try:
        F.set_gemm_precision('bf16')
        tr = Trainer(opt, m, CtdetLoss(opt, consts).to(dev), lr=1e-2)
        tr.train_step(batch, 0)
        o = tr.optimizer
        n = o.n_live
        assert torch.equal(o.flat_p16[:n], o.flat_p[:n].to(torch.bfloat16))
        F.set_gemm_precision('fp32')
        tr.train_step(batch, 0)                                  # fp32 step: master weights move, the bf16 copy does not
        assert not torch.equal(o.flat_p16[:n], o.flat_p[:n].to(torch.bfloat16))
        assert not o._p16_synced
        F.set_gemm_precision('bf16')
        assert all(F.shadow_of(p) is None for p in o.params)     # the stale views are off the parameters
        plain = run()                                            # converts the fp32 master weights while staging
        old, F.BF16_SHADOWS = F.BF16_SHADOWS, False
        try:
            want = run()
        finally:
            F.BF16_SHADOWS = old
        assert torch.equal(plain[0]['verts3d']['left'], want[0]['verts3d']['left'])
        ev = tr.evaluation([batch])                              # refreshes and re-attaches
        assert np.isfinite(ev['mpjpe_mm']) and o._p16_synced
        assert torch.equal(o.flat_p16[:n], o.flat_p[:n].to(torch.bfloat16))
        again = run()
        assert torch.equal(again[0]['verts3d']['left'], want[0]['verts3d']['left'])
        tr.train_step(batch, 0)
        assert torch.equal(o.flat_p16[:n], o.flat_p[:n].to(torch.bfloat16))
    finally:
        F.set_gemm_precision('fp32')
