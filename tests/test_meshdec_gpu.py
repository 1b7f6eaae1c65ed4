"""Fused mesh decoder (csrc/meshdec.hip): one DualGraphLayer per (hand, sample) workgroup against the unfused kernel chain -- which is itself pinned to
the reference's module-level goldens (tests/test_modules_gpu.py) and to the oracle end to end."""
import numpy as np
import pytest
import torch

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


def _layer(level, seed=0, drop=0.05):
    from pdfnet_amd.networks import intaghand_decoder as D
    torch.manual_seed(seed)
    g = D.load_graph_constants()
    V = (63, 126, 252)[level]
    cin, cout = (512, 256, 128)[level], (256, 128, 64)[level]
    layer = D.DualGraphLayer(V, cin, cout, g['ell_left'][level], g['ell_right'][level], 4, [12, 24, 48][level], 256, (256, 128, 64)[level], 4, drop)
    for n, p in layer.named_parameters():                  # biases / LayerNorm parameters off their init values (zeros / ones hide indexing errors)
        if p.dim() == 1:
            p.data.add_(0.1 * torch.randn_like(p))
    return layer.cuda()


@pytest.mark.parametrize("level", [0, 1, 2])
@pytest.mark.parametrize("B", [3, 32])
def test_fused_level_forward_equals_the_unfused_chain_in_eval_mode(level, B):
    from pdfnet_amd import functional as F
    layer = _layer(level).eval()
    V, cin = (63, 126, 252)[level], (512, 256, 128)[level]
    x = torch.randn(2, B, V, cin, generator=torch.Generator().manual_seed(level + 10 * B)).cuda()
    with torch.no_grad():
        xin = x + layer.position_embeddings.weight
        blocks = list(zip(layer.graph_left.GCN_blocks, layer.graph_right.GCN_blocks))
        from pdfnet_amd.networks.intaghand_decoder import gcn_block_pair
        ref = xin
        for i, (bl, br) in enumerate(blocks):
            ref = gcn_block_pair(bl, br, ref, relu_out=i != 3)
        ref = layer.attn(ref)
        out, _, _, _ = F.mesh_level_forward(layer, xin, training=False)
    torch.cuda.synchronize()
    err = float((out - ref).abs().max())
    scale = float(ref.abs().max())
    print("level %d B %d: max |fused - unfused| = %.2e (max |ref| %.2f)" % (level, B, err, scale))
    assert torch.isfinite(out).all()
    assert err <= 2e-5 * max(1.0, scale)


@pytest.mark.parametrize("level", [0, 1, 2])
def test_fused_level_forward_draws_the_same_dropout_masks_as_the_unfused_chain(level):
    """Train mode with dropout ON: the fused kernels hash (seed, element index) exactly like the unfused dropout / LayerNorm / attention kernels,
    and the seeds are drawn in the same order -- so the two paths agree to rounding, mask for mask."""
    from pdfnet_amd import functional as F
    from pdfnet_amd.networks.intaghand_decoder import gcn_block_pair
    B = 4
    layer = _layer(level, seed=3, drop=0.3).train()
    V, cin = (63, 126, 252)[level], (512, 256, 128)[level]
    x = torch.randn(2, B, V, cin, generator=torch.Generator().manual_seed(level)).cuda()
    with torch.no_grad():
        xin = x + layer.position_embeddings.weight
        F.manual_seed(77)
        ref = xin
        for i, (bl, br) in enumerate(zip(layer.graph_left.GCN_blocks, layer.graph_right.GCN_blocks)):
            ref = gcn_block_pair(bl, br, ref, relu_out=i != 3)
        ref = layer.attn(ref)
        F.manual_seed(77)
        out, _, _, _ = F.mesh_level_forward(layer, xin, training=True)
    torch.cuda.synchronize()
    err = float((out - ref).abs().max())
    print("level %d train, p = 0.3: max |fused - unfused| = %.2e (max |ref| %.2f)" % (level, err, float(ref.abs().max())))
    assert err <= 5e-5 * max(1.0, float(ref.abs().max()))


def _unfused(layer, x):
    from pdfnet_amd.networks.intaghand_decoder import gcn_block_pair
    h = x + layer.position_embeddings.weight
    for i, (bl, br) in enumerate(zip(layer.graph_left.GCN_blocks, layer.graph_right.GCN_blocks)):
        h = gcn_block_pair(bl, br, h, relu_out=i != 3)
    return layer.attn(h)


@pytest.mark.parametrize("drop", [0.0, 0.3])
@pytest.mark.parametrize("level", [0, 1, 2])
def test_fused_level_backward_equals_the_unfused_chain(level, drop):
    """dx and every parameter gradient of one DualGraphLayer: the fused backward (five launches + the weight-gradient GEMMs on the side stream)
    against autograd through the unfused kernels, same seeds (so with dropout on, the same masks)."""
    from pdfnet_amd import functional as F
    B = 5
    layer = _layer(level, seed=5, drop=drop).train()
    V, cin = (63, 126, 252)[level], (512, 256, 128)[level]
    x0 = torch.randn(2, B, V, cin, generator=torch.Generator().manual_seed(level + 3)).cuda()
    gy = torch.randn(2, B, V, cin // 2, generator=torch.Generator().manual_seed(level + 4)).cuda()
    def both(gy):
        res = {}
        for fused in (False, True):
            layer.zero_grad(set_to_none=True)
            x = x0.clone().requires_grad_()
            F.manual_seed(99)
            F.MESH_FUSED = fused
            try:
                out = layer(x) if fused else _unfused(layer, x)
            finally:
                F.MESH_FUSED = True
            out.backward(gy)
            F.join_wgrad()
            torch.cuda.synchronize()
            res[fused] = (out.detach().clone(), x.grad.clone(), {n: p.grad.clone() for n, p in layer.named_parameters() if p.grad is not None})
        return res[False], res[True]
    (o0, dx0, g0), (o1, dx1, g1) = both(gy)
    assert float((o0 - o1).abs().max()) <= 5e-5 * max(1.0, float(o0.abs().max()))
    # Two correct fp32 implementations round LayerNorm differently, so a ReLU input within ~1e-7 of zero can fall on opposite sides in the two
    # forwards (one element per ~10^6): that (hand, sample)'s data gradient then differs by a few per cent.  Such a sample is a COUNTED exception
    # (at most one, printed): its sample's output gradient is zeroed and both paths run again, so that every other sample and EVERY parameter gradient is
    # held to the tight bar -- no loose bar anywhere.
    top = float(dx0.abs().max())
    per = (dx0 - dx1).abs().amax((2, 3)) / top                              # [2, B]
    flipped = int((per > 1e-4).sum())
    assert flipped <= 1 and float(per.max()) <= 0.2, per
    if flipped:
        h, b = [int(v) for v in (per > 1e-4).nonzero()[0]]
        print("level %d p %.1f: ReLU flip in (hand %d, sample %d), data gradient off by %.1e of the largest: excluded and re-run" % (level, drop, h, b, float(per.max())))
        gy2 = gy.clone()
        gy2[:, b] = 0                                                       # both hands of the sample: they exchange keys / values in the cross-hand attention
        (o0, dx0, g0), (o1, dx1, g1) = both(gy2)
        top = float(dx0.abs().max())
        per = (dx0 - dx1).abs().amax((2, 3)) / top
    assert int((per > 1e-4).sum()) == 0, per
    assert set(g0) == set(g1), set(g0) ^ set(g1)
    worst = []
    for n in g0:
        topg = float(g0[n].abs().max())
        if n.endswith('w_ks.bias'):                                         # exactly zero in exact arithmetic (softmax is shift-invariant): rounding noise in both
            topg = float(g0[n[:-4] + 'weight'].abs().max())
        err = float((g0[n] - g1[n]).abs().max())
        worst.append((err / (topg + 1e-30), n))
        assert err <= 2e-4 * topg + 1e-6, (n, err, topg)
    worst.sort(reverse=True)
    print("level %d p %.1f: %d parameter gradients, %d excluded ReLU-flip sample(s), worst relative deviations %s" % (level, drop, len(g0), flipped, ["%s %.1e" % (n, e) for e, n in worst[:3]]))


@pytest.mark.parametrize("level", [0, 1, 2])
def test_bf16_build_of_the_fused_level_is_as_good_a_bf16_path_as_the_per_op_chain(level):
    """csrc/meshdec_bf16.hip: the same kernels with their LINEAR products on the bf16 MFMA (operands rounded to bf16, fp32 accumulation; what the
    library's bf16 mode does to every linear layer of the per-op chain).  One DualGraphLayer, forward + backward, dropout off, three ways: fp32
    fused (the reference here), bf16 fused, bf16 per-op.  bf16 rounding moves ReLU inputs across zero, so the two bf16 paths sit ~0.5 % (output),
    ~11 % (input gradient), ~6 % (weight gradients) from fp32 in relative L2 -- BOTH of them; the fused build must be no further from fp32 than
    the per-op chain (x 1.15) and the two bf16 paths must be closer to each other than either is to fp32."""
    from pdfnet_amd import functional as F
    B = 5
    layer = _layer(level, seed=6, drop=0.0).train()
    V, cin = (63, 126, 252)[level], (512, 256, 128)[level]
    x0 = torch.randn(2, B, V, cin, generator=torch.Generator().manual_seed(level + 5)).cuda()
    gy = torch.randn(2, B, V, cin // 2, generator=torch.Generator().manual_seed(level + 6)).cuda()
    res = {}
    try:
        for name, bf, fused in (("fp32", False, True), ("bf16 fused", True, True), ("bf16 per-op", True, False)):
            F.set_gemm_precision('bf16' if bf else 'fp32')
            F.MESH_FUSED = fused
            layer.zero_grad(set_to_none=True)
            x = x0.clone().requires_grad_()
            F.manual_seed(99)
            out = layer(x)
            out.backward(gy)
            F.join_wgrad()
            torch.cuda.synchronize()
            res[name] = (out.detach().clone(), x.grad.clone(), torch.cat([p.grad.flatten() for n, p in layer.named_parameters() if p.grad is not None and p.dim() > 1]))
    finally:
        F.set_gemm_precision('fp32')
        F.MESH_FUSED = True

    def l2(a, b):
        return float((a - b).norm() / a.norm())
    for k, name in enumerate(("out", "dx", "dW")):
        ref, fu, po = res["fp32"][k], res["bf16 fused"][k], res["bf16 per-op"][k]
        d_fu, d_po, d_between = l2(ref, fu), l2(ref, po), l2(po, fu)
        print("level %d %s: bf16 fused vs fp32 %.2e, bf16 per-op vs fp32 %.2e, fused vs per-op %.2e (relative L2)" % (level, name, d_fu, d_po, d_between))
        assert torch.isfinite(fu).all() and d_fu > 1e-5                        # the bf16 path really ran
        assert d_fu <= 1.15 * d_po + 1e-4, (name, d_fu, d_po)
        assert d_between <= 0.6 * max(d_fu, d_po), (name, d_between, d_fu, d_po)
