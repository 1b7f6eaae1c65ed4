"""Fused mesh decoder (csrc/meshdec.hip): one DualGraphLayer per (hand, sample) workgroup against the unfused kernel chain -- which is itself pinned to
the reference's module-level goldens (tests/test_modules_gpu.py) and to the oracle end to end."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


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
