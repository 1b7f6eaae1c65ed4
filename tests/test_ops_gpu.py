"""GPU parity of every C-ABI kernel (called through pdfnet_amd.functional -> ctypes -> libpdfnet_hip.so)
against a plain PyTorch fp32 CPU reference of the same op (or the oracle for the point / MANO ops)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as TF

from tests.util import gold, T

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def F():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from pdfnet_amd import functional as F
    return F


def dev(t):
    return t.cuda()


def close(a, b, atol, rtol=1e-5, what=""):
    a = a.detach().cpu().double()
    b = b.detach().cpu().double()
    err = (a - b).abs().max().item()
    lim = atol + rtol * b.abs().max().item()
    assert err <= lim, "%s: max err %.3e > %.3e (max|ref|=%.3e)" % (what, err, lim, b.abs().max().item())


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


@pytest.mark.parametrize("M,K,N,act,bias", [
    (2016, 512, 256, 0, True), (300, 131, 128, 1, True), (64, 252, 778, 0, False), (8064, 64, 3, 0, True),
    (1000, 16, 64, 2, True), (37, 1024, 509, 0, True), (40000, 2, 128, 0, False), (33000, 42, 130, 2, True), (70000, 16, 64, 1, True), (32768, 64, 64, 1, False), (40000, 256, 128, 0, True), (5, 3, 3, 2, True),
    # few output tiles under a long reduction: split-K (partials through the scratch ring + splitk_finish), odd widths included
    (64, 4608, 1024, 0, False), (64, 1024, 1024, 0, True), (32, 1024, 512, 1, True), (192, 252, 778, 2, True), (2048, 2048, 128, 1, True),
    (100, 777, 130, 0, True)])
def test_linear(F, M, K, N, act, bias):
    x, w, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5), rnd(N, seed=3) if bias else None
    xr, wr = x.clone().requires_grad_(), w.clone().requires_grad_()
    br = b.clone().requires_grad_() if bias else None
    ref = TF.linear(xr, wr, br)
    ref = TF.relu(ref) if act == 1 else (TF.leaky_relu(ref, 0.1) if act == 2 else ref)
    gy = rnd(M, N, seed=4)
    ref.backward(gy)
    xd, wd = dev(x).requires_grad_(), dev(w).requires_grad_()
    bd = dev(b).requires_grad_() if bias else None
    out = F.linear(xd, wd, bd, act)
    out.backward(dev(gy))
    tol = 2e-5 * max(1.0, K ** 0.5 / 8)
    close(out, ref, tol, what="linear fwd")
    close(xd.grad, xr.grad, tol * 4, what="linear dx")
    close(wd.grad, wr.grad, 2e-5 * max(1.0, M ** 0.5 / 4), rtol=2e-5, what="linear dw")
    if bias:
        close(bd.grad, br.grad, 2e-5 * max(1.0, M ** 0.5 / 4), rtol=2e-5, what="linear db")


CONVS = [  # N, Cin, H, W, Cout, k, stride, pad, act, bias
    (2, 3, 32, 32, 64, 7, 2, 3, 0, False), (2, 3, 17, 19, 3, 3, 1, 1, 1, False), (2, 64, 16, 16, 64, 3, 1, 1, 0, False),
    (2, 128, 13, 13, 128, 3, 2, 1, 0, False), (2, 256, 16, 16, 128, 1, 2, 0, 0, False), (2, 64, 16, 16, 256, 1, 1, 0, 0, False),
    (3, 32, 9, 11, 48, 3, 1, 1, 1, True), (1, 256, 64, 64, 256, 3, 1, 1, 1, True), (2, 256, 8, 8, 122, 1, 1, 0, 0, True),
    (4, 1024, 16, 16, 256, 3, 1, 1, 0, False),
    # streaming kernels for tiny channel counts: e_conv1 (3 -> 3, full resolution), the 2-channel hm / mask heads
    (2, 3, 192, 200, 3, 3, 1, 1, 1, False), (2, 256, 64, 64, 2, 1, 1, 0, 0, True), (2, 128, 128, 128, 2, 1, 1, 0, 0, True),
    # >= 600 128x128 tiles of a 3x3 stride-1 conv with >= 256 input channels: the LDS-halo kernel, W = 64 / 32 / 16 (no ReLU here: with 10 M
    # outputs a few pre-activations sit within rounding of 0 and flip the mask between implementations)
    (5, 256, 64, 64, 512, 3, 1, 1, 0, True), (5, 512, 64, 64, 256, 3, 1, 1, 0, False), (20, 272, 32, 32, 512, 3, 1, 1, 0, False),
    (20, 512, 32, 32, 256, 3, 1, 1, 0, True), (80, 256, 16, 16, 512, 3, 1, 1, 0, True), (80, 512, 16, 16, 256, 3, 1, 1, 0, False),
    # valid 3x3 on the 5x5 / 3x3 centre windows (few rows, long reduction: 32x32 tiles)
    (64, 256, 5, 5, 512, 3, 1, 0, 0, False), (64, 512, 3, 3, 1024, 3, 1, 0, 0, False),
    # the ResNet stem at output widths that are multiples of 64: its dedicated MFMA weight-gradient kernel (image borders on all sides)
    (3, 3, 128, 128, 64, 7, 2, 3, 0, False), (2, 3, 64, 256, 64, 7, 2, 3, 1, False),
    # the LDS-DMA weight-gradient kernel's issue paths: scalar offsets (16-pixel K-steps: stride 2, a 48-wide map that wraps every
    # third step, plain 1x1 rows) and the per-row form (a 40-wide map)
    (24, 128, 64, 64, 256, 3, 2, 1, 0, False), (12, 128, 48, 48, 256, 3, 1, 1, 0, False), (24, 256, 32, 32, 512, 1, 1, 0, 0, True),
    (16, 128, 40, 40, 256, 3, 1, 1, 0, False)]


@pytest.mark.parametrize("cfg", CONVS)
def test_conv2d(F, cfg):
    N, Cin, H, W, Cout, k, s, p, act, bias = cfg
    x = rnd(N, Cin, H, W, seed=1)
    w = rnd(Cout, Cin, k, k, seed=2, scale=(Cin * k * k) ** -0.5)
    b = rnd(Cout, seed=3) if bias else None
    xr, wr = x.clone().requires_grad_(), w.clone().requires_grad_()
    br = b.clone().requires_grad_() if bias else None
    ref = TF.conv2d(xr, wr, br, s, p)
    if act:
        ref = TF.relu(ref)
    gy = rnd(*ref.shape, seed=4)
    ref.backward(gy)
    xd = dev(x).contiguous(memory_format=torch.channels_last).requires_grad_()
    wd = dev(w).contiguous(memory_format=torch.channels_last).requires_grad_()
    bd = dev(b).requires_grad_() if bias else None
    out = F.conv2d(xd, wd, bd, s, p, act)
    assert out.shape == ref.shape
    out.backward(dev(gy))
    K = Cin * k * k
    close(out, ref, 3e-5 * max(1, K ** 0.5 / 16), what="conv fwd")
    close(xd.grad, xr.grad, 1e-4, rtol=2e-5, what="conv dx")
    close(wd.grad, wr.grad, 5e-5 * max(1, (N * H * W) ** 0.5 / 16), rtol=5e-5, what="conv dw")
    if bias:
        close(bd.grad, br.grad, 1e-4, rtol=5e-5, what="conv db")


@pytest.mark.parametrize("cfg", [(2, 32, 8, 8, 16, 4, 2, 1), (2, 6, 5, 7, 10, 4, 2, 1), (2, 64, 4, 4, 32, 4, 4, 0),
                                 (2, 128, 2, 2, 256, 8, 8, 0), (3, 512, 32, 32, 256, 4, 2, 1)])
def test_deconv2d(F, cfg):
    N, Cin, H, W, Cout, k, s, p = cfg
    x = rnd(N, Cin, H, W, seed=1)
    w = rnd(Cin, Cout, k, k, seed=2, scale=Cin ** -0.5)
    b = rnd(Cout, seed=3)
    xr, wr, br = x.clone().requires_grad_(), w.clone().requires_grad_(), b.clone().requires_grad_()
    ref = TF.conv_transpose2d(xr, wr, br, s, p)
    gy = rnd(*ref.shape, seed=4)
    ref.backward(gy)
    xd = dev(x).contiguous(memory_format=torch.channels_last).requires_grad_()
    wd = dev(w).contiguous(memory_format=torch.channels_last).requires_grad_()
    bd = dev(b).requires_grad_()
    out = F.deconv2d(xd, wd, bd, s, p)
    assert out.shape == ref.shape
    out.backward(dev(gy))
    close(out, ref, 5e-5, what="deconv fwd")
    close(xd.grad, xr.grad, 2e-4, rtol=2e-5, what="deconv dx")
    close(wd.grad, wr.grad, 2e-4, rtol=5e-5, what="deconv dw")
    close(bd.grad, br.grad, 2e-4, rtol=5e-5, what="deconv db")


@pytest.mark.parametrize("shape,relu,res", [((4, 64, 9, 9), True, False), ((2, 256, 8, 8), True, True), ((3, 3, 5, 5), False, False),
                                            ((700, 130), True, False), ((2, 128, 31, 33), False, True)])
def test_batchnorm(F, shape, relu, res):
    C = shape[1]
    x = rnd(*shape, seed=1) * 2 + 0.5
    r = rnd(*shape, seed=5) if res else None
    g, b = torch.rand(C) + 0.5, rnd(C, seed=2)
    rm0, rv0 = rnd(C, seed=3), torch.rand(C) + 0.5
    for training in (True, False):
        xr, gr, br = x.clone().requires_grad_(), g.clone().requires_grad_(), b.clone().requires_grad_()
        rr = r.clone().requires_grad_() if res else None
        rm, rv = rm0.clone(), rv0.clone()
        ref = TF.batch_norm(xr, rm, rv, gr, br, training, 0.1, 1e-5)
        if res:
            ref = ref + rr
        if relu:
            ref = TF.relu(ref)
        xd, gd, bd = dev(x).requires_grad_(), dev(g).requires_grad_(), dev(b).requires_grad_()
        if len(shape) == 4:
            xd = dev(x).contiguous(memory_format=torch.channels_last).requires_grad_()
        rd = dev(r).requires_grad_() if res else None
        rmd, rvd = dev(rm0.clone()), dev(rv0.clone())
        out = F.batch_norm(xd, gd, bd, rmd, rvd, training, 0.1, 1e-5, relu, rd)
        close(out, ref, 2e-5, what="bn fwd train=%s" % training)
        if training:
            close(rmd, rm, 1e-5, what="running_mean")
            close(rvd, rv, 1e-5, what="running_var")
            gy = rnd(*shape, seed=4)
            ref.backward(gy)
            out.backward(dev(gy))
            close(xd.grad, xr.grad, 5e-5, rtol=5e-5, what="bn dx")
            close(gd.grad, gr.grad, 2e-4, rtol=5e-5, what="bn dgamma")
            close(bd.grad, br.grad, 2e-4, rtol=5e-5, what="bn dbeta")
            if res:
                close(rd.grad, rr.grad, 1e-6, what="bn dres")


@pytest.mark.parametrize("R,K,C", [(96, 64, 128), (40, 128, 1024), (300, 64, 256), (7, 5, 8)])
def test_fused_batchnorm_relu_max_over_neighbours(F, R, K, C):
    """Set-abstraction tail (intaghand_encoder.py:59-62): BatchNorm2d -> ReLU -> MaxPool over K in one pass, forward and
    backward (gradient of the convolution output, dgamma, dbeta), train and eval mode, against the unfused torch ops."""
    x = rnd(R * K, C, seed=1) * 1.5 + 0.3
    g, b = torch.rand(C) + 0.5, rnd(C, seed=2) * 0.5
    g[::7] *= -1.0                                           # negative scales: the max then sits on the smallest input
    rm0, rv0 = rnd(C, seed=3), torch.rand(C) + 0.5
    for training in (True, False):
        xr, gr, br = x.clone().requires_grad_(), g.clone().requires_grad_(), b.clone().requires_grad_()
        rm, rv = rm0.clone(), rv0.clone()
        z = TF.relu(TF.batch_norm(xr, rm, rv, gr, br, training, 0.1, 1e-5))
        ref = z.view(R, K, C).max(1)[0]
        xd, gd, bd = dev(x).requires_grad_(), dev(g).requires_grad_(), dev(b).requires_grad_()
        rmd, rvd = dev(rm0.clone()), dev(rv0.clone())
        out = F.bn_relu_max_over_k(xd, gd, bd, rmd, rvd, K, training, 0.1, 1e-5)
        close(out, ref, 2e-5, what="fused bn-relu-max fwd train=%s" % training)
        if training:
            close(rmd, rm, 1e-5, what="running_mean")
            close(rvd, rv, 1e-5, what="running_var")
            gy = rnd(R, C, seed=4)
            ref.backward(gy)
            out.backward(dev(gy))
            close(xd.grad, xr.grad, 5e-5, rtol=5e-5, what="fused dx")
            close(gd.grad, gr.grad, 2e-4, rtol=5e-5, what="fused dgamma")
            close(bd.grad, br.grad, 2e-4, rtol=5e-5, what="fused dbeta")


@pytest.mark.parametrize("N,S,K,Cin,Cout", [(1024, 512, 64, 3, 64), (512, 128, 64, 131, 128)])
def test_group_then_conv_equals_conv_then_gather(F, N, S, K, Cin, Cout):
    """group_points(_2) + the set-abstraction MLP's first 1x1 convolution (lib/utils/utils.py:134-188 +
    intaghand_encoder.py:48-52,67-71) evaluated without the grouped tensor: conv(p)[idx] - W[:, :3] centre must equal the
    convolution of the gathered, centre-subtracted block -- values and the gradients wrt points, weight and bias."""
    B = 3
    pad = (Cin + 15) // 16 * 16
    g = torch.Generator().manual_seed(N + Cin)
    pts = torch.cat([torch.rand(B, N, 2, generator=g) * 0.2 - 0.1, torch.rand(B, N, 1, generator=g) * 0.1 + 0.4,
                     torch.randn(B, N, Cin - 3, generator=g)], 2)
    w = torch.randn(Cout, Cin, generator=g) * Cin ** -0.5
    b = torch.randn(Cout, generator=g)
    rows = dev(torch.nn.functional.pad(pts, (0, pad - Cin))).requires_grad_()
    wd, bd = dev(torch.nn.functional.pad(w, (0, pad - Cin))).requires_grad_(), dev(b).requires_grad_()
    idx = F.knn_ball_indices(rows, S, K, 0.0225 if Cin == 3 else 0.04)
    u = F.linear(rows, wd, bd)
    ctr = torch.nn.functional.pad(rows[:, :S, :3], (0, pad - 3))
    y = F.gather_sub(u, F.linear(ctr, wd), idx)
    # reference: gather, subtract the centre from xyz, convolve (float64 on the CPU)
    pr, wr, br = pts.double().requires_grad_(), w.double().requires_grad_(), b.double().requires_grad_()
    ii = idx.cpu().long()
    grouped = torch.gather(pr.unsqueeze(1).expand(B, S, N, Cin), 2, ii.unsqueeze(-1).expand(B, S, K, Cin))
    sub = torch.cat([pr[:, :S, None, :3], torch.zeros(B, S, 1, Cin - 3, dtype=torch.float64)], 3)
    ref = (grouped - sub) @ wr.t() + br
    close(y, ref.float(), 2e-5, rtol=2e-5, what="gather_sub fwd")
    gy = torch.randn(ref.shape, generator=g)
    ref.backward(gy.double())
    y.backward(dev(gy))
    close(rows.grad[..., :Cin], pr.grad.float(), 2e-4, rtol=5e-5, what="d points")
    close(wd.grad[:, :Cin], wr.grad.float(), 2e-3, rtol=1e-4, what="d weight")
    close(bd.grad, br.grad.float(), 2e-3, rtol=1e-4, what="d bias")


def test_gather_sub_backward_is_deterministic_and_needs_no_atomics(F):
    """pdf_invert_index + pdf_gather_sub_bwd_sorted: every point's slot list is the ascending list of (centroid, neighbour) slots
    that picked it; the backward is bit-identical from run to run and equals the atomic form to summation-order accuracy -- incl.
    points nobody picked (zero rows) and a hub point picked by every centroid."""
    B, N, S, K, C = 3, 512, 128, 64, 128
    g = torch.Generator().manual_seed(5)
    idx = torch.randint(0, N, (B, S, K), generator=g, dtype=torch.int32)
    idx[0, :, 0] = 7                                          # a hub: 128 slots
    idx[1][idx[1] == 11] = 12                                 # point 11 of cloud 1 is picked by nobody
    u, v = dev(torch.randn(B, N, C, generator=g)).requires_grad_(), dev(torch.randn(B, S, C, generator=g)).requires_grad_()
    gy = dev(torch.randn(B, S, K, C, generator=g))
    idd = dev(idx)
    L = F._L()
    start = torch.empty((B, N + 1), dtype=torch.int32, device='cuda')
    lst = torch.empty((B, S * K), dtype=torch.int32, device='cuda')
    L.pdf_invert_index(idd.data_ptr(), B, N, S * K, start.data_ptr(), lst.data_ptr(), None, None)
    torch.cuda.synchronize()
    flat = idx.reshape(B, -1).long()
    for b in range(B):
        order = torch.sort(flat[b], stable=True)[1].int()      # slots grouped by point, ascending inside a group
        assert torch.equal(lst[b].cpu(), order)
        assert torch.equal(start[b].cpu().long(), torch.cat([torch.zeros(1, dtype=torch.long), torch.bincount(flat[b], minlength=N).cumsum(0)]))
    # degenerate clouds (an invalid hand's all-zero cloud: every centroid picks the same few points) and shapes that do not divide into
    # whole 64-slot rounds per wave, incl. a point count that leaves LDS room for fewer than 16 per-wave histograms
    for (B2, N2, E2, hot) in ((2, 1024, 512 * 64, 64), (2, 37, 1000, 3), (1, 4000, 5000, 4000), (1, 1, 130, 1)):
        i2 = torch.randint(0, hot, (B2, E2), generator=g, dtype=torch.int32)
        i2[0, ::7] = N2 - 1
        s2 = torch.empty((B2, N2 + 1), dtype=torch.int32, device='cuda')
        l2 = torch.empty((B2, E2), dtype=torch.int32, device='cuda')
        L.pdf_invert_index(dev(i2).data_ptr(), B2, N2, E2, s2.data_ptr(), l2.data_ptr(), None, None)
        torch.cuda.synchronize()
        for b in range(B2):
            assert torch.equal(l2[b].cpu(), torch.sort(i2[b].long(), stable=True)[1].int()), (N2, E2, hot)
            assert torch.equal(s2[b].cpu().long(), torch.cat([torch.zeros(1, dtype=torch.long), torch.bincount(i2[b].long(), minlength=N2).cumsum(0)]))
    outs = []
    for mode in (True, True, False):
        F.GATHER_SORTED = mode
        try:
            uu, vv = u.detach().clone().requires_grad_(), v.detach().clone().requires_grad_()
            F.gather_sub(uu, vv, idd).backward(gy)
            torch.cuda.synchronize()
            outs.append((uu.grad.clone(), vv.grad.clone()))
        finally:
            F.GATHER_SORTED = True
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])          # run to run: identical bits
    close(outs[0][0], outs[2][0], 1e-4, rtol=1e-5, what="du sorted vs atomic")
    close(outs[0][1], outs[2][1], 1e-4, rtol=1e-5, what="dv")
    assert float(outs[0][0][1, 11].abs().max()) == 0.0
    ref = torch.zeros(B, N, C, dtype=torch.float64).index_put_((torch.arange(B)[:, None].expand(B, S * K).reshape(-1), flat.reshape(-1)),
                                                                gy.cpu().double().reshape(-1, C), accumulate=True)
    close(outs[0][0], ref.float(), 1e-4, rtol=1e-5, what="du vs float64")


def test_pool_upsample_relu(F):
    x = rnd(2, 16, 13, 14, seed=1)
    xr = x.clone().requires_grad_()
    ref = TF.max_pool2d(xr, 3, 2, 1)
    gy = rnd(*ref.shape, seed=2)
    ref.backward(gy)
    xd = dev(x).requires_grad_()
    out = F.maxpool3s2(xd)
    out.backward(dev(gy))
    close(out, ref, 0, what="maxpool")
    close(xd.grad, xr.grad, 1e-6, what="maxpool dx")
    xr = x.clone().requires_grad_()
    ref = TF.interpolate(xr, scale_factor=2, mode='bilinear', align_corners=True)
    gy = rnd(*ref.shape, seed=3)
    ref.backward(gy)
    xd = dev(x).requires_grad_()
    out = F.upsample2x(xd)
    out.backward(dev(gy))
    close(out, ref, 2e-6, what="upsample")
    close(xd.grad, xr.grad, 1e-5, what="upsample dx")
    for shape in ((3, 2, 16, 16), (2, 3, 5, 7)):            # channel counts the float4 kernels decline (the 2-channel mask head): gather form
        x2 = rnd(*shape, seed=4)
        xr = x2.clone().requires_grad_()
        ref = TF.interpolate(xr, scale_factor=2, mode='bilinear', align_corners=True)
        gy2 = rnd(*ref.shape, seed=5)
        ref.backward(gy2)
        xd = dev(x2).requires_grad_()
        out = F.upsample2x(xd)
        out.backward(dev(gy2))
        close(out, ref, 2e-6, what="upsample C=%d" % shape[1])
        close(xd.grad, xr.grad, 1e-5, what="upsample dx C=%d" % shape[1])
    xd = dev(x).requires_grad_()
    out = F.relu(xd)
    out.backward(dev(gy[:, :, :13, :14].contiguous()))
    close(out, TF.relu(x), 0, what="relu")
    close(xd.grad, gy[:, :, :13, :14] * (x > 0), 0, what="relu dx")


def test_pad2d_pads_and_crops_in_one_launch(F):
    """F.pad2d (pdf_pad2d): zero-padding of a matrix to [rows, cols] and, as its backward, the crop of the gradient -- against
    torch.nn.functional.pad; incl. a single row (a bias), a 3-column cloud and the no-op."""
    for (R, C, R2, C2) in ((131, 259, 144, 272), (1, 131, 1, 144), (2048, 3, 2048, 16), (64, 64, 64, 64), (5, 7, 9, 7)):
        x = rnd(R, C, seed=R + C)
        xr = x.clone().requires_grad_()
        ref = TF.pad(xr, (0, C2 - C, 0, R2 - R))
        g = rnd(R2, C2, seed=3)
        ref.backward(g)
        xd = dev(x).requires_grad_()
        out = F.pad2d(xd, R2, C2)
        out.backward(dev(g))
        close(out, ref, 0, what="pad2d %s" % ((R, C, R2, C2),))
        close(xd.grad, xr.grad, 0, what="pad2d dx")


def test_l2norm_layernorm(F):
    x = rnd(2, 256, 7, 9, seed=1)
    w = torch.rand(256) * 10 + 5
    xr, wr = x.clone().requires_grad_(), w.clone().requires_grad_()
    n = xr.pow(2).sum(1, keepdim=True).sqrt() + 1e-10
    ref = wr.view(1, -1, 1, 1) * (xr / n)
    gy = rnd(*x.shape, seed=2)
    ref.backward(gy)
    xd, wd = dev(x).requires_grad_(), dev(w).requires_grad_()
    out = F.l2norm(xd, wd)
    out.backward(dev(gy))
    close(out, ref, 1e-5, what="l2norm")
    close(xd.grad, xr.grad, 1e-5, rtol=1e-5, what="l2norm dx")
    close(wd.grad, wr.grad, 2e-5, rtol=1e-5, what="l2norm dw")
    for Fd in (509, 512, 64, 16):
        x = rnd(3, 63, Fd, seed=3) * 3 + 1
        g, b = torch.rand(Fd) + 0.5, rnd(Fd, seed=4)
        xr, gr, br = x.clone().requires_grad_(), g.clone().requires_grad_(), b.clone().requires_grad_()
        ref = TF.layer_norm(xr, (Fd,), gr, br, 1e-6)
        gy = rnd(*x.shape, seed=5)
        ref.backward(gy)
        xd, gd, bd = dev(x).requires_grad_(), dev(g).requires_grad_(), dev(b).requires_grad_()
        out = F.layer_norm(xd, gd, bd, 1e-6)
        out.backward(dev(gy))
        close(out, ref, 1e-5, what="ln")
        close(xd.grad, xr.grad, 2e-5, rtol=1e-5, what="ln dx")
        close(gd.grad, gr.grad, 5e-5, rtol=1e-5, what="ln dgamma")
        close(bd.grad, br.grad, 5e-5, rtol=1e-5, what="ln dbeta")


def _check_group(F, pts_rows, C, S, K, r2, ldg, gold_idx_sorted, ref_grouped):
    """pts_rows [B,N,C] (xyz first). ref_grouped [B,C,S,K] from the oracle."""
    from oracle import pdfnet_cpu as O
    pd = dev(pts_rows).requires_grad_()
    g, idx = F.knn_ball_group(pd, C, S, K, r2, ldg)
    idx_s = np.sort(idx.cpu().numpy(), -1)
    # exact-tie handling: rows whose K-th and (K+1)-th smallest distances are equal may pick either
    xyz = pts_rows[:, :, :3]
    d2 = ((xyz.unsqueeze(1) - xyz[:, :S].unsqueeze(2)) ** 2)
    d2 = d2[..., 0] + d2[..., 1] + d2[..., 2]
    srt = torch.sort(d2, dim=2)[0]
    tie = (srt[:, :, K - 1] == srt[:, :, K]).numpy() if d2.shape[2] > K else np.zeros(idx_s.shape[:2], bool)
    same = (idx_s == gold_idx_sorted).all(-1)
    assert same[~tie].all(), "index sets differ on %d tie-free centroids" % int((~same[~tie]).sum())
    # grouped features: permutation-invariant comparison (sum / max over K)
    # (on exact ties -- duplicated points -- either duplicate may be picked: there only xyz is comparable)
    gg = g[..., :C].permute(0, 3, 1, 2).detach().cpu()
    nt = torch.from_numpy(~tie).unsqueeze(1)                                    # [B,1,S]
    ds, dm = (gg.sum(-1) - ref_grouped.sum(-1)).abs(), (gg.abs().amax(-1) - ref_grouped.abs().amax(-1)).abs()
    assert (ds * nt).max() <= 1e-4 and (dm * nt).max() == 0
    assert ds[:, :3].max() <= 1e-4 and dm[:, :3].max() == 0
    if ldg > C:
        assert (g[..., C:] == 0).all()
    return pd, g, idx


def test_knn_ball_group_level1(F):
    from oracle import pdfnet_cpu as O
    gd = gold("op_group_points_l1")
    pts = T(gd["points"])
    ref, _ = O.group_level1(pts, 512, 64, float(gd["r2"]))
    pd, g, idx = _check_group(F, pts, 3, 512, 64, float(gd["r2"]), 16, gd["idx_sorted"].astype(np.int64), ref)
    # backward vs autograd of the oracle gather with the kernel's own indices
    gy = rnd(*g.shape, seed=3)
    gy[..., 3:] = 0
    g.backward(dev(gy))
    pr = pts.clone().requires_grad_()
    ii = idx.cpu().long()
    gath = torch.gather(pr, 1, ii.reshape(3, -1, 1).expand(-1, -1, 3)).view(3, 512, 64, 3) - pr[:, :512].unsqueeze(2)
    (gath * gy[..., :3]).sum().backward()
    close(pd.grad, pr.grad, 1e-4, what="group bwd")


@pytest.mark.parametrize("N,S", [(700, 350), (100, 100), (1000, 512), (65, 17)])
def test_knn_ball_group_takes_any_cloud_size(F, N, S):
    """lib/utils/utils.py:134-163 has no restriction on opt.SAMPLE_NUM: N need not be 64 * 2^k.  Index sets against the
    oracle on a cloud with far outliers (so the ball mask fires) for ragged N."""
    from oracle import pdfnet_cpu as O
    K, r2 = 64, 0.015
    g = torch.Generator().manual_seed(N)
    pts = torch.cat([(torch.rand(3, N, 2, generator=g) - 0.5) * 0.2, 0.4 + 0.1 * torch.rand(3, N, 1, generator=g)], -1)
    far = torch.rand(3, N, generator=g) < 0.1
    pts[..., :2] = torch.where(far.unsqueeze(-1), (torch.rand(3, N, 2, generator=g) - 0.5), pts[..., :2])
    ref, _ = O.group_level1(pts, S, K, r2)
    gold_idx = np.sort(O.knn_ball_indices(pts, S, K, r2).numpy(), -1)
    _check_group(F, pts, 3, S, K, r2, 16, gold_idx, ref)


def test_knn_ball_group_level2(F):
    from oracle import pdfnet_cpu as O
    gd = gold("op_group_points_l2")
    feat = T(gd["feat"])                                   # [B,131,512]
    ref, _ = O.group_level2(feat, 128, 64, float(gd["r2"]))
    rows = feat.transpose(1, 2).contiguous()
    pd, g, idx = _check_group(F, rows, 131, 128, 64, float(gd["r2"]), 144, gd["idx_sorted"].astype(np.int64), ref)
    gy = rnd(*g.shape, seed=3)
    g.backward(dev(gy))
    pr = rows.clone().requires_grad_()
    ii = idx.cpu().long()
    gath = torch.gather(pr, 1, ii.reshape(3, -1, 1).expand(-1, -1, 131)).view(3, 128, 64, 131)
    gath = torch.cat([gath[..., :3] - pr[:, :128, :3].unsqueeze(2), gath[..., 3:]], -1)
    (gath * gy[..., :131]).sum().backward()
    close(pd.grad, pr.grad, 2e-4, what="group2 bwd")


def test_gather_rows_maxk_sft(F):
    from oracle import pdfnet_cpu as O
    gd = gold("op_gather_feat")
    out = F.gather_rows(dev(T(gd["feat"])), dev(T(gd["ind"])))
    close(out, T(gd["out"]), 0, what="gather golden")
    feat = rnd(2, 64, 32, 32, seed=1)
    R = 64
    choose = torch.randint(0, R * R, (2, 100))
    fr = feat.clone().requires_grad_()
    c2 = (choose // R // 2) * (R // 2) + choose % R // 2
    ref = O.gather_hw(fr, c2[:, :40])
    gy = rnd(2, 40, 64, seed=2)
    ref.backward(gy)
    fd = dev(feat).requires_grad_()
    out = F.gather_rows(fd, dev(choose)[:, :40], R=R, shift=1)
    out.backward(dev(gy))
    close(out, ref, 0, what="gather pyramid")
    close(fd.grad, fr.grad, 1e-5, what="gather bwd")
    x = rnd(50, 64, 24, seed=3)
    xr = x.clone().requires_grad_()
    ref = xr.max(1)[0]
    gy = rnd(50, 24, seed=4)
    ref.backward(gy)
    xd = dev(x).requires_grad_()
    out = F.max_over_k(xd)
    out.backward(dev(gy))
    close(out, ref, 0, what="maxk")
    close(xd.grad, xr.grad, 0, what="maxk bwd")
    a, b, c = rnd(7, 9, 13, seed=5), rnd(7, 9, 13, seed=6), rnd(7, 9, 13, seed=7)
    ar, br, cr = (t.clone().requires_grad_() for t in (a, b, c))
    ref = ar * (br + 1) + cr
    gy = rnd(7, 9, 13, seed=8)
    ref.backward(gy)
    ad, bd, cd = (dev(t).requires_grad_() for t in (a, b, c))
    out = F.sft_modulate(ad, bd, cd)
    out.backward(dev(gy))
    close(out, ref, 1e-6, what="sft")
    for u, v in ((ad, ar), (bd, br), (cd, cr)):
        close(u.grad, v.grad, 1e-6, what="sft bwd")


def _ell(D):
    V = D.shape[0]
    Wd = int((D != 0).sum(1).max())
    col = torch.zeros(V, Wd, dtype=torch.int32)
    val = torch.zeros(V, Wd)
    for v in range(V):
        nz = torch.nonzero(D[v]).flatten()
        col[v, :len(nz)] = nz.int()
        val[v, :len(nz)] = D[v, nz]
    return col, val


def test_sft3_layer_in_one_kernel_equals_the_four_convs(F):
    """SFTLayer(3, 3) (sft0, reference intaghand_encoder.py:205-219): fused kernel vs the layer written out in torch --
    output, d fea, d cond and the eight parameter gradients."""
    R = 5000
    fea, cond, gy = rnd(2, R // 2, 3, seed=1), rnd(2, R // 2, 3, seed=2), rnd(2, R // 2, 3, seed=3)
    params = [rnd(3, 3, 1, 1, seed=10 + i, scale=0.6) if i % 2 == 0 else rnd(3, seed=10 + i, scale=0.3) for i in range(8)]
    fr, cr = fea.clone().requires_grad_(), cond.clone().requires_grad_()
    pr = [p.clone().requires_grad_() for p in params]

    def conv(x, w, b):
        return x @ w.flatten(1).t() + b
    scale = conv(TF.leaky_relu(conv(cr, pr[0], pr[1]), 0.1), pr[2], pr[3])
    shift = conv(TF.leaky_relu(conv(cr, pr[4], pr[5]), 0.1), pr[6], pr[7])
    ref = fr * (scale + 1) + shift
    ref.backward(gy)
    fd, cd = dev(fea).requires_grad_(), dev(cond).requires_grad_()
    pd = [dev(p).requires_grad_() for p in params]
    out = F.sft3(fd, cd, pd)
    out.backward(dev(gy))
    close(out, ref, 2e-6, what="sft3 fwd")
    close(fd.grad, fr.grad, 2e-6, what="sft3 dfea")
    close(cd.grad, cr.grad, 5e-6, what="sft3 dcond")
    for i, (a, b) in enumerate(zip(pd, pr)):
        close(a.grad, b.grad, 2e-4, rtol=2e-5, what="sft3 dparam %d" % i)


def test_conv_with_skip_accumulates_the_shortcut_gradient_in_the_epilogue(F):
    """ResNet identity block input: conv1's backward-data adds onto the shortcut's gradient (pdf_conv2d_bwd_data_add) -- same dx
    and dw as the plain graph, with a channels_last shortcut gradient (in place) and a strided one (fallback add)."""
    x, w = rnd(3, 64, 12, 10, seed=1), rnd(32, 64, 1, 1, seed=2, scale=0.1)
    a = rnd(3, 32, 12, 10, seed=3)
    for b in (rnd(3, 64, 12, 10, seed=4).contiguous(memory_format=torch.channels_last), rnd(3, 64, 12, 10, seed=5)):
        xr, wr = dev(x).requires_grad_(), dev(w).requires_grad_()
        ((F.conv2d(xr, wr) * dev(a)).sum() + (xr * dev(b)).sum()).backward()
        xd, wd = dev(x).requires_grad_(), dev(w).requires_grad_()
        y, sc = F.conv2d_with_skip(xd, wd)
        ((y * dev(a)).sum() + (sc * dev(b)).sum()).backward()
        close(xd.grad, xr.grad, 1e-6, what="skip dx")
        close(wd.grad, wr.grad, 1e-6, what="skip dw")


def test_l2norm_cat_writes_the_pyramid_in_place(F):
    """F.l2norm_cat == torch.cat([F.l2norm(x_i, w_i)], 1), forward and all gradients."""
    Cs = (16, 32, 8, 24)
    xs = [rnd(2, C, 6, 5, seed=10 + i) for i, C in enumerate(Cs)]
    ws = [rnd(C, seed=20 + i).abs() + 0.5 for i, C in enumerate(Cs)]
    gy = rnd(2, sum(Cs), 6, 5, seed=30)
    a = [dev(x).requires_grad_() for x in xs] + [dev(w).requires_grad_() for w in ws]
    b = [dev(x).requires_grad_() for x in xs] + [dev(w).requires_grad_() for w in ws]
    ref = torch.cat([F.l2norm(x, w) for x, w in zip(a[:4], a[4:])], 1)
    ref.backward(dev(gy))
    out = F.l2norm_cat(b[:4], b[4:])
    out.backward(dev(gy))
    close(out, ref, 0.0, rtol=0.0, what="l2norm_cat fwd")
    for u, v in zip(b, a):
        close(u.grad, v.grad, 1e-6, what="l2norm_cat grad")


def test_cheby_attention(F):
    from oracle import pdfnet_cpu as O
    Ls = O.load_graph_constants()['L_right']
    for D, Fd in zip(Ls, (32, 16, 8)):
        V = D.shape[0]
        col, val = _ell(D)
        colT, valT = _ell(D.t().contiguous())
        x = rnd(3, V, Fd, seed=1)
        xr = x.clone().requires_grad_()
        ref = torch.stack((xr, torch.einsum('vw,bwf->bvf', D, xr)), -1).flatten(2)
        gy = rnd(3, V, 2 * Fd, seed=2)
        ref.backward(gy)
        xd = dev(x).requires_grad_()
        out = F.cheby2(xd, tuple(dev(t) for t in (col, val, colT, valT)))
        out.backward(dev(gy))
        close(out, ref, 1e-5, what="cheby")
        close(xd.grad, xr.grad, 1e-5, what="cheby bwd")
    for V, Fd in ((63, 256), (126, 128), (252, 64), (63, 16)):
        q, k, v = rnd(2, V, Fd, seed=3), rnd(2, V, Fd, seed=4), rnd(2, V, Fd, seed=5)
        qr, kr, vr = (t.clone().requires_grad_() for t in (q, k, v))
        ref = O.mha(qr, kr, vr, 4, lambda a: a)
        gy = rnd(2, V, Fd, seed=6)
        ref.backward(gy)
        qd, kd, vd = (dev(t).requires_grad_() for t in (q, k, v))
        out = F.attention(qd, kd, vd, 4)
        out.backward(dev(gy))
        close(out, ref, 2e-5, what="attn")
        for u, w_ in ((qd, qr), (kd, kr), (vd, vr)):
            close(u.grad, w_.grad, 5e-5, rtol=2e-5, what="attn bwd")
    # dropout: keep-rate and backward uses the same mask
    x = dev(torch.ones(1 << 20)).requires_grad_()
    y = F.dropout(x, 0.05, True)
    keep = (y > 0).float().mean().item()
    assert abs(keep - 0.95) < 2e-3 and abs(y.max().item() - 1 / 0.95) < 1e-6
    y.sum().backward()
    assert torch.equal(x.grad > 0, y > 0)
    qd = dev(q).requires_grad_()
    o1 = F.attention(qd, dev(k), dev(v), 4, 0.3, True)
    assert torch.isfinite(o1).all()


def test_mano_regressor(F):
    from oracle import pdfnet_cpu as O
    from oracle import synth
    gd = gold("op_mano_layer")
    for side in ("left", "right"):
        c = synth.synthetic_mano_consts(side)
        cd = {k: dev(v.contiguous()) for k, v in c.items()}
        v, j = F.mano_lbs(cd, dev(T(gd[side + "_rot"])), dev(T(gd[side + "_pose"])), dev(T(gd[side + "_shape"])),
                          trans=dev(T(gd[side + "_trans"])), side=side)
        close(v, T(gd[side + "_verts"]), 5e-6, what="mano verts " + side)
        close(j, T(gd[side + "_joints"]), 5e-6, what="mano joints " + side)
        reg = O.full_regressor(c['J_regressor'])
        vd = v.clone().requires_grad_()
        jr = F.regress_joints(dev(reg), vd)
        close(jr, T(gd[side + "_full_regressor_joints"]), 5e-6, what="regressor")
        gy = rnd(*jr.shape, seed=1)
        jr.backward(dev(gy))
        close(vd.grad, torch.einsum('jv,bjc->bvc', reg, gy), 1e-6, what="regressor bwd")
        v2, j2 = F.mano_lbs(cd, dev(T(gd[side + "_rot"])), dev(T(gd[side + "_pose"])), dev(T(gd[side + "_shape"])), side=side, center_idx=9)
        vo, jo = O.mano_lbs(c, T(gd[side + "_rot"]), T(gd[side + "_pose"]), T(gd[side + "_shape"]), side=side, center_idx=9)
        close(v2, vo, 5e-6, what="mano centered")
        close(j2, jo, 5e-6, what="mano centered joints")


@pytest.mark.parametrize("side,center", [("left", None), ("right", 9)])
def test_mano_lbs_backward_against_the_oracle_autograd(F, side, center):
    """pdf_mano_lbs_bwd: gradients of (verts, joints) w.r.t. root / pose axis-angles, shape and translation against autograd
    through the float64 oracle restatement of ManoLayer.forward (manolayer.py:257-334) -- incl. a zero rotation (|a| = 0)."""
    from oracle import pdfnet_cpu as O
    from oracle import synth
    c = synth.synthetic_mano_consts(side)
    c64 = {k: v.double() for k, v in c.items()}
    g = torch.Generator().manual_seed(11)
    B = 4
    rot, pose = torch.randn(B, 3, generator=g) * 0.8, torch.randn(B, 45, generator=g) * 0.4
    pose[0, 6:9] = 0.0                                       # an exactly-zero joint rotation
    shape, trans = torch.randn(B, 10, generator=g), torch.randn(B, 3, generator=g) * 0.1
    gv, gj = torch.randn(B, 778, 3, generator=g), torch.randn(B, 21, 3, generator=g)
    leaves = [t.double().requires_grad_() for t in (rot, pose, shape, trans)]
    # the oracle helper builds float32 identity matrices: run it in float64 by default dtype
    old = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    try:
        vo, jo = O.mano_lbs(c64, leaves[0], leaves[1], leaves[2], trans=leaves[3], side=side, center_idx=center)
        ((vo * gv.double()).sum() + (jo * gj.double()).sum()).backward()
    finally:
        torch.set_default_dtype(old)
    cd = {k: dev(v.contiguous()) for k, v in c.items()}
    dl = [dev(t).requires_grad_() for t in (rot, pose, shape, trans)]
    v, j = F.mano_lbs(cd, dl[0], dl[1], dl[2], trans=dl[3], side=side, center_idx=center)
    close(v, vo.float(), 1e-5, what="mano verts")
    ((v * dev(gv)).sum() + (j * dev(gj)).sum()).backward()
    for name, a, b in zip(("d root", "d pose", "d shape", "d trans"), dl, leaves):
        close(a.grad, b.grad.float(), 1e-4, rtol=2e-4, what=name)
    # only one output used, shape / trans not wanted
    d2 = [dev(t).requires_grad_() for t in (rot, pose)]
    v2, _ = F.mano_lbs(cd, d2[0], d2[1], dev(shape), side=side, center_idx=center)
    (v2 * dev(gv)).sum().backward()
    l2 = [t.double().requires_grad_() for t in (rot, pose)]
    torch.set_default_dtype(torch.float64)
    try:
        vo2, _ = O.mano_lbs(c64, l2[0], l2[1], shape.double(), side=side, center_idx=center)
        (vo2 * gv.double()).sum().backward()
    finally:
        torch.set_default_dtype(old)
    close(d2[0].grad, l2[0].grad.float(), 1e-4, rtol=2e-4, what="d root (verts only)")
    close(d2[1].grad, l2[1].grad.float(), 1e-4, rtol=2e-4, what="d pose (verts only)")


def test_mano_split_coeff_decode_and_fix_shape(F):
    """ManoRender.Split_coeff (Mano_render.py:145-194): the decode of the 122-channel params head that feeds the MANO layers in
    the reference's legacy branch (simplified.py:730-736), values and gradient; `fix_shape` (interhand.py:120-123)."""
    from oracle import synth
    from pdfnet_amd.utils import fix_shape, mano_from_params
    B, R, down = 3, 64, 4
    g = torch.Generator().manual_seed(5)
    P = torch.randn(B, 122, R // down, R // down, generator=g)
    ind = torch.tensor([[0, 255], [17, 200], [100, 3]])
    K = torch.tensor([[[70.0, 0, 30], [0, 66, 34], [0, 0, 1]]]).repeat(B, 1, 1) + torch.rand(B, 3, 3, generator=g)
    Pr = P.clone().requires_grad_()
    gsz = R // down
    ref = {}
    for h in range(2):                                       # the reference's formulas, per hand
        th = Pr.reshape(B, 122, -1)[torch.arange(B), :, ind[:, h]]
        t = th[:, 61 * h + 58:61 * h + 61]
        tz = t[:, 2] + 0.6
        cx, cy = (ind[:, h] % gsz) * down, (ind[:, h] // gsz) * down
        ref[h] = (th[:, 61 * h:61 * h + 3], th[:, 61 * h + 3:61 * h + 48], th[:, 61 * h + 48:61 * h + 58] * 0,
                  torch.stack((tz * (t[:, 0] + cx - K[:, 0, 2]) / K[:, 0, 0], tz * (t[:, 1] + cy - K[:, 1, 2]) / K[:, 1, 1], tz), 1))
    Pd = dev(P).contiguous(memory_format=torch.channels_last).requires_grad_()
    o, p, s, t = F.mano_split_coeff(Pd, dev(ind), dev(K), R, down)
    for h in range(2):
        for a, b, w in zip((o[h], p[h], s[h], t[h]), ref[h], ("orient", "pose", "shape", "trans")):
            close(a, b.detach(), 1e-6, rtol=1e-6, what="split_coeff " + w)
    go, gp, gt = torch.randn(2, B, 3, generator=g), torch.randn(2, B, 45, generator=g), torch.randn(2, B, 3, generator=g)
    ((o * dev(go)).sum() + (p * dev(gp)).sum() + (t * dev(gt)).sum()).backward()
    sum((ref[h][0] * go[h]).sum() + (ref[h][1] * gp[h]).sum() + (ref[h][3] * gt[h]).sum() for h in range(2)).backward()
    close(Pd.grad, Pr.grad, 1e-6, rtol=1e-5, what="split_coeff bwd")
    # the decoded parameters drive both MANO layers
    consts = {h: {k: dev(v.contiguous()) for k, v in synth.synthetic_mano_consts(h).items()} for h in ("left", "right")}
    v, j, tr = mano_from_params(consts, Pd.detach(), dev(ind), dev(K), R, down)
    assert v.shape == (2, B, 778, 3) and j.shape == (2, B, 21, 3) and torch.isfinite(v).all()
    # fix_shape: flips the left x-rows exactly when left and right agree there
    c2 = {h: {k: v.clone() for k, v in consts[h].items()} for h in consts}
    c2['left']['shapedirs'][:, 0, :] = c2['right']['shapedirs'][:, 0, :]
    before = c2['left']['shapedirs'].clone()
    assert fix_shape(c2)
    assert torch.equal(c2['left']['shapedirs'][:, 0, :], -before[:, 0, :]) and torch.equal(c2['left']['shapedirs'][:, 1:, :], before[:, 1:, :])
    assert not fix_shape(c2)                                 # now they differ: left alone


def test_adam(F):
    from pdfnet_amd import hip
    p0, g = rnd(10000, seed=1), rnd(10000, seed=2)
    pr = p0.clone().requires_grad_()
    opt = torch.optim.Adam([pr], lr=1e-4)
    pd, m, v = dev(p0.clone()), dev(torch.zeros(10000)), dev(torch.zeros(10000))
    for step in range(1, 4):
        pr.grad = g * step
        opt.step()
        corr = dev(torch.tensor([1 - 0.9 ** step, 1 - 0.999 ** step]))
        hip.lib().pdf_adam_step(hip.ptr(pd), hip.ptr(dev(g * step)), hip.ptr(m), hip.ptr(v), 10000, 1e-4, 0.9, 0.999, 1e-8,
                                hip.ptr(corr), 1.0, hip.stream())
    close(pd, pr, 1e-7, what="adam")


def test_nms_top1_decode(F):
    from oracle import pdfnet_cpu as O
    gd = gold("op_nms_topk")
    ind, score = F.nms_top1(dev(T(gd["hm"])))
    assert np.array_equal(ind.cpu().numpy(), gd["ind"])                       # reference _nms + _topk, bit-exact
    for seed, (B, H, W) in enumerate([(4, 64, 64), (2, 96, 96), (3, 7, 5)]):
        hm = rnd(B, 2, H, W, seed=seed + 20)
        ind, score = F.nms_top1(dev(hm))
        assert torch.equal(ind.cpu(), O.nms_topk_center(hm))
        assert torch.equal(score.cpu(), hm.reshape(B, 2, -1).gather(2, ind.cpu().unsqueeze(-1)).squeeze(-1))


def test_mano_gt_from_coeff(F):
    from oracle import pdfnet_cpu as O
    from oracle import synth
    from pdfnet_amd.utils import mano_gt_from_coeff
    g = torch.Generator().manual_seed(3)
    coeff = torch.randn(5, 124, generator=g) * 0.3
    K = torch.tensor([[[256.0, 0, 128], [0, 256, 128], [0, 0, 1]]]).repeat(5, 1, 1)
    consts = {h: synth.synthetic_mano_consts(h) for h in ("left", "right")}
    out = mano_gt_from_coeff({h: {k: dev(v.contiguous()) for k, v in c.items()} for h, c in consts.items()}, dev(coeff), dev(K))
    for hi, hand in enumerate(("left", "right")):
        p = coeff[:, 62 * hi:62 * (hi + 1)]
        v, j = O.mano_lbs(consts[hand], p[:, 4:7], p[:, 7:52], p[:, 52:62], trans=p[:, 1:4], side=hand)
        close(out[hand]['verts3d'], v, 5e-6, what="gt verts")
        close(out[hand]['joints3d'], j, 5e-6, what="gt joints")
        pj = j @ K.transpose(1, 2)
        close(out[hand]['joints2d'], pj[..., :2] / pj[..., 2:], 1e-3, rtol=1e-5, what="gt joints2d")


def test_depth2pcl_front_end(F):
    """Device depth->cloud front end vs the (reference-pinned) oracle: same candidate sets, the wrap-pad multiset is
    exact, the >1024 case draws a uniformly random 1024-subset, clouds are the back-projected pixels."""
    from oracle import pdfnet_cpu as O
    g = gold("op_depth2pcl")
    depth, mask, K = g["depth"], g["mask"], g["K"]
    B = 3
    dd = dev(T(depth))[None, None].repeat(B, 1, 1, 1)
    mm = dev(T(mask))[None].repeat(B, 1, 1, 1)
    KK = dev(T(K))[None].repeat(B, 1, 1)
    valid = dev(torch.tensor([[1.0, 1.0], [1.0, 0.0], [1.0, 1.0]]))
    choose, cloud, count = F.depth2pcl(dd, mm, KK, valid, seed=123)
    choose2, _, _ = F.depth2pcl(dd, mm, KK, valid, seed=124)
    ptsL, candL = O.depth_candidates(depth, mask[1], K)
    ptsR, candR = O.depth_candidates(depth, mask[0], K)
    ch = choose.cpu().numpy()
    assert count.cpu().numpy()[0].tolist() == [len(candL), len(candR)]
    # left: n <= 1024 -> np.pad(..., 'wrap') multiset, random order
    want = np.sort(np.pad(candL, (0, 1024 - len(candL)), 'wrap'))
    for b in range(B):
        assert np.array_equal(np.sort(ch[b, 0]), want)
    assert not np.array_equal(ch[0, 0], np.sort(ch[0, 0]))                       # shuffled
    # right: n > 1024 -> 1024 distinct candidates
    for b in (0, 2):
        sel = ch[b, 1]
        assert len(np.unique(sel)) == 1024 and np.isin(sel, candR).all()
    rank = np.searchsorted(candR, ch[0, 1])
    assert abs(rank.mean() / len(candR) - 0.5) < 0.05                            # uniform over the candidates
    assert not np.array_equal(np.sort(ch[0, 1]), np.sort(choose2.cpu().numpy()[0, 1]))   # seed changes the subset
    assert not np.array_equal(ch[0, 1], ch[2, 1])                                # per-(sample, hand) streams differ
    # invalid hand: zero indices and a zero cloud
    assert (ch[1, 1] == 0).all() and (cloud[1, 1] == 0).all()
    # clouds = back-projected pixels
    close(cloud[0, 0], T(ptsL.T[ch[0, 0]]), 1e-6, rtol=1e-5, what="left cloud")
    close(cloud[0, 1], T(ptsR.T[ch[0, 1]]), 1e-6, rtol=1e-5, what="right cloud")
    # degenerate: nothing inside the mask -> all-zero indices
    c0, cl0, n0 = F.depth2pcl(dd, mm * 0, KK, valid, seed=1)
    assert (c0 == 0).all() and (n0 == 0).all()


def test_paired_decoder_ops(F):
    """The paired / fused launches of the mesh decoder against their one-branch-at-a-time compositions (torch CPU)."""
    from oracle import pdfnet_cpu as O
    B, V, K, N = 3, 63, 128, 64
    # linear_pair (+ReLU) = two Linear layers
    x = rnd(2, B, V, K, seed=1)
    ws = [rnd(N, K, seed=2 + i, scale=K ** -0.5) for i in range(2)]
    bs = [rnd(N, seed=4 + i) for i in range(2)]
    gy = rnd(2, B, V, N, seed=6)
    xr = x.clone().requires_grad_()
    wr = [w.clone().requires_grad_() for w in ws]
    br = [b.clone().requires_grad_() for b in bs]
    ref = torch.stack([TF.relu(TF.linear(xr[i], wr[i], br[i])) for i in range(2)])
    ref.backward(gy)
    xd = dev(x).requires_grad_()
    wd = [dev(w).requires_grad_() for w in ws]
    bd = [dev(b).requires_grad_() for b in bs]
    out = F.linear_pair(xd, wd[0], bd[0], wd[1], bd[1], F.ACT_RELU)
    out.backward(dev(gy))
    close(out, ref, 3e-5, what="linear_pair")
    close(xd.grad, xr.grad, 1e-4, what="linear_pair dx")
    for i in range(2):
        close(wd[i].grad, wr[i].grad, 1e-4, rtol=2e-5, what="linear_pair dw%d" % i)
        close(bd[i].grad, br[i].grad, 1e-4, rtol=2e-5, what="linear_pair db%d" % i)

    # layer_norm_fused: paired parameters, ReLU, residual add (no dropout), extra gradient on the residual stream
    Fd = 96
    x, a = rnd(2, B, V, Fd, seed=7), rnd(2, B, V, Fd, seed=8)
    gs = [rnd(Fd, seed=9 + i) for i in range(2)]
    be = [rnd(Fd, seed=11 + i) for i in range(2)]
    gz, gy = rnd(2, B, V, Fd, seed=13), rnd(2, B, V, Fd, seed=14)
    xr, ar = x.clone().requires_grad_(), a.clone().requires_grad_()
    gr = [t.clone().requires_grad_() for t in gs]
    er = [t.clone().requires_grad_() for t in be]
    zr = xr + ar
    yr = torch.stack([TF.relu(TF.layer_norm(zr[i], (Fd,), gr[i], er[i], 1e-6)) for i in range(2)])
    ((zr * gz).sum() + (yr * gy).sum()).backward()
    xd, ad = dev(x).requires_grad_(), dev(a).requires_grad_()
    gd = [dev(t).requires_grad_() for t in gs]
    ed = [dev(t).requires_grad_() for t in be]
    zd, yd = F.layer_norm_fused(xd, gd[0], ed[0], 1e-6, F.ACT_RELU, add=ad, p=0.5, training=False, gamma1=gd[1], beta1=ed[1])
    ((zd * dev(gz)).sum() + (yd * dev(gy)).sum()).backward()
    close(zd, zr, 1e-6, what="ln_fused z")
    close(yd, yr, 2e-5, what="ln_fused y")
    close(xd.grad, xr.grad, 5e-5, what="ln_fused dx")
    close(ad.grad, ar.grad, 5e-5, what="ln_fused dadd")
    for i in range(2):
        close(gd[i].grad, gr[i].grad, 2e-4, rtol=2e-5, what="ln_fused dgamma%d" % i)
        close(ed[i].grad, er[i].grad, 2e-4, rtol=2e-5, what="ln_fused dbeta%d" % i)
    # single parameter set, no add, no act == plain layer_norm
    y1 = F.layer_norm_fused(dev(x), gd[0], ed[0])
    close(y1, TF.layer_norm(x, (Fd,), gs[0], be[0], 1e-6), 2e-5, what="ln_fused plain")
    # dropout inside the fused op: the dropped operand's gradient carries the forward mask, the residual's does not
    xd, ad = dev(torch.zeros(2, B, V, Fd)).requires_grad_(), dev(torch.ones(2, B, V, Fd)).requires_grad_()
    zd, yd = F.layer_norm_fused(xd, gd[0], ed[0], add=ad, p=0.25, training=True, gamma1=gd[1], beta1=ed[1])
    keep = zd > 0
    assert abs(keep.float().mean().item() - 0.75) < 0.02 and abs(zd.max().item() - 1 / 0.75) < 1e-6
    zd.sum().backward()
    assert torch.equal(ad.grad > 0, keep) and torch.equal(xd.grad, torch.ones_like(xd))
    # dropout_add
    xd, rd = dev(torch.ones(1 << 16)).requires_grad_(), dev(torch.zeros(1 << 16)).requires_grad_()
    yd = F.dropout_add(xd, rd, 0.1, True)
    yd.sum().backward()
    assert torch.equal(xd.grad > 0, yd > 0) and torch.equal(rd.grad, torch.ones_like(rd))
    close(F.dropout_add(xd, rd + 2, 0.1, False), torch.full((1 << 16,), 3.0), 0, what="dropout_add eval")

    # cheby2_pair = left and right Laplacians
    g = O.load_graph_constants()
    DL, DR = g['L_left'][0], g['L_right'][0]
    ells = []
    for D in (DL, DR):
        col, val = _ell(D)
        colT, valT = _ell(D.t().contiguous())
        ells.append([col, val, colT, valT])
    for k in range(4):                                                   # common ELL width
        w = max(ells[0][k].shape[1], ells[1][k].shape[1])
        for e in ells:
            e[k] = TF.pad(e[k], (0, w - e[k].shape[1]))
    x = rnd(2, B, 63, 32, seed=15)
    gy = rnd(2, B, 63, 64, seed=16)
    xr = x.clone().requires_grad_()
    ref = torch.stack([torch.stack((xr[i], torch.einsum('vw,bwf->bvf', D, xr[i])), -1).flatten(2) for i, D in enumerate((DL, DR))])
    ref.backward(gy)
    xd = dev(x).requires_grad_()
    out = F.cheby2_pair(xd, tuple(dev(t) for t in ells[0]), tuple(dev(t) for t in ells[1]))
    out.backward(dev(gy))
    close(out, ref, 1e-5, what="cheby2_pair")
    close(xd.grad, xr.grad, 1e-5, what="cheby2_pair bwd")

    # attention with kv_shift = B: each hand's queries against the other hand's keys / values
    q, k, v = rnd(2, B, V, 64, seed=17), rnd(2, B, V, 64, seed=18), rnd(2, B, V, 64, seed=19)
    gy = rnd(2, B, V, 64, seed=20)
    qr, kr, vr = (t.clone().requires_grad_() for t in (q, k, v))
    ref = torch.stack((O.mha(qr[0], kr[1], vr[1], 4, lambda a: a), O.mha(qr[1], kr[0], vr[0], 4, lambda a: a)))
    ref.backward(gy)
    qd, kd, vd = (dev(t).requires_grad_() for t in (q, k, v))
    out = F.attention(qd, kd, vd, 4, kv_shift=B)
    out.backward(dev(gy))
    close(out, ref, 2e-5, what="attn kv_shift")
    for u, w_ in ((qd, qr), (kd, kr), (vd, vr)):
        close(u.grad, w_.grad, 5e-5, rtol=2e-5, what="attn kv_shift bwd")


def test_mesh_loss_kernels(F):
    """rowloss / face_loss against the torch composition of lib/trains/simplified.py:66-115 and F.l1_loss / F.mse_loss."""
    G, B, V = 2, 3, 778
    pred, gt = rnd(G, B, V, 3, seed=1, scale=0.05), rnd(G, B, V, 3, seed=2, scale=0.05)
    gen = torch.Generator().manual_seed(3)
    faces = torch.stack([torch.stack([torch.randperm(V, generator=gen)[:3] for _ in range(1538)]) for _ in range(G)])
    unit = lambda v: TF.normalize(v, p=2, dim=2)

    def ref_terms(p, q, fc):
        f0, f1, f2 = fc[:, 0], fc[:, 1], fc[:, 2]
        n = unit(torch.cross(unit(q[:, f1] - q[:, f0]), unit(q[:, f2] - q[:, f0]), dim=2))
        cos = [torch.abs((unit(v) * n).sum(2, keepdim=True)) for v in (p[:, f1] - p[:, f0], p[:, f2] - p[:, f0], p[:, f2] - p[:, f1])]
        d = lambda x, i, j: torch.sqrt(((x[:, i] - x[:, j]) ** 2).sum(2, keepdim=True))
        ed = [torch.abs(d(p, i, j) - d(q, i, j)) for i, j in ((f0, f1), (f0, f2), (f1, f2))]
        return torch.cat(cos, 1).mean(), torch.cat(ed, 1).mean()
    pr = pred.clone().requires_grad_()
    refs = [ref_terms(pr[g], gt[g], faces[g]) for g in range(G)]
    wn, we = torch.tensor([1.5, -0.7]), torch.tensor([0.3, 2.0])
    sum(wn[g] * refs[g][0] + we[g] * refs[g][1] for g in range(G)).backward()
    pd = dev(pred).requires_grad_()
    nl, el = F.face_loss(pd, dev(gt), dev(faces))
    (nl * dev(wn) + el * dev(we)).sum().backward()
    close(nl, torch.stack([r[0] for r in refs]), 1e-6, rtol=2e-5, what="normal loss")
    close(el, torch.stack([r[1] for r in refs]), 1e-7, rtol=2e-5, what="edge loss")
    close(pd.grad, pr.grad, 1e-7, rtol=2e-4, what="face loss grad")
    # without the edge gradient (alpha == 0 in the trainer): only the normal term reaches pred
    pr.grad = None
    refs = [ref_terms(pr[g], gt[g], faces[g]) for g in range(G)]
    sum(wn[g] * refs[g][0] for g in range(G)).backward()
    pd.grad = None
    nl, el = F.face_loss(pd, dev(gt), dev(faces), edge_grad=False)
    (nl * dev(wn) + el * 0.0).sum().backward()
    close(pd.grad, pr.grad, 1e-7, rtol=2e-4, what="face loss grad (normal only)")

    for mode, rd in (('l1', 2), ('l2', 1), ('l1', 1)):
        a, b = rnd(2, 5, 21, 3, seed=4), rnd(2, 5, 21, 3, seed=5)
        b[0, 0, 0, 0] = a[0, 0, 0, 0]                                   # an exact tie: sign(0) = 0 like torch
        ar = a.clone().requires_grad_()
        e = (ar - b).abs() if mode == 'l1' else (ar - b) ** 2
        ref = e.reshape(*a.shape[:rd], -1).mean(-1)
        w = rnd(*a.shape[:rd], seed=6)
        (ref * w).sum().backward()
        ad = dev(a).requires_grad_()
        out = F.rowloss(ad, dev(b), rd, mode)
        (out * dev(w)).sum().backward()
        close(out, ref, 1e-6, what="rowloss " + mode)
        close(ad.grad, ar.grad, 1e-7, rtol=1e-5, what="rowloss grad " + mode)


def test_fps(F):
    """pdf_fps picks, in order, against the oracle restatement of the reference helper (bit-exact indices)."""
    from oracle import pdfnet_cpu as O
    g = gold("op_fps")
    for name in ("a", "b", "dup"):
        pts, S, start = g["pts_" + name], int(g["S_" + name][0]), int(g["start_" + name][0])
        ref = O.fps_order(pts, S, start)
        got = F.fps(dev(torch.from_numpy(pts))[None], S, dev(torch.tensor([start], dtype=torch.int32)))
        assert np.array_equal(got[0].cpu().numpy().astype(np.int64), ref), name
        assert np.array_equal(np.unique(got[0].cpu().numpy()), g["unique_" + name]), name
    # batched, padded rows, start defaulting to 0, a large cloud
    gen = torch.Generator().manual_seed(5)
    big = torch.rand(3, 9000, 4, generator=gen)
    got = F.fps(dev(big), 64).cpu().numpy()
    for b in range(3):
        assert np.array_equal(got[b].astype(np.int64), O.fps_order(big[b, :, :3].numpy(), 64, 0)), b



def test_fps_single_wave_kernel_and_the_reorder_option(F):
    """Clouds of <= 1,024 points take the single-wave kernel (16 points per lane, DPP arg-max): picks bit-exact against the oracle
    for ragged sizes, duplicated points (frozen distances) and explicit first picks; `fps_reorder` -- the reference's
    `--sample_strategy FPS` (lib/opts.py:231, the commented block lib/datasets/interhand.py:857-900) -- against its restatement."""
    from oracle import pdfnet_cpu as O
    gen = torch.Generator().manual_seed(11)
    for N, S in ((1024, 512), (1000, 128), (65, 64), (300, 300), (64, 7)):
        pts = torch.rand(5, N, 3, generator=gen)
        pts[1, N // 2:] = pts[1, :N - N // 2].clone()             # duplicates: their distances freeze at <= 1e-8
        start = torch.randint(0, N, (5,), generator=gen, dtype=torch.int32)
        got = F.fps(dev(pts), S, dev(start)).cpu().numpy()
        for b in range(5):
            assert np.array_equal(got[b].astype(np.int64), O.fps_order(pts[b].numpy(), S, int(start[b]))), (N, S, b)
    pts = torch.rand(6, 1024, 3, generator=gen)
    pts[2, 700:] = pts[2, :324].clone()
    choose = torch.randint(0, 65536, (6, 1024), generator=gen)
    s1 = torch.randint(0, 1024, (6,), generator=gen, dtype=torch.int32)
    s2 = torch.randint(0, 512, (6,), generator=gen, dtype=torch.int32)
    c, ch = F.fps_reorder(dev(pts), dev(choose), 512, 128, dev(s1), dev(s2))
    for b in range(6):
        rc, rch = O.fps_reorder(pts[b].numpy(), choose[b].numpy(), 512, 128, int(s1[b]), int(s2[b]))
        assert np.array_equal(ch[b].cpu().numpy(), rch), b
        assert np.array_equal(c[b].cpu().numpy(), rc), b
        assert sorted(ch[b].cpu().tolist()) == sorted(choose[b].tolist())        # a permutation of the drawn points


def test_limits_and_empty_inputs(F):
    """Error behaviour at the documented limits (RuntimeError, not a crash or a silent fallback) and empty inputs."""
    with pytest.raises(RuntimeError):
        F.layer_norm(dev(torch.zeros(4, 2048)), dev(torch.ones(2048)), dev(torch.zeros(2048)))      # F <= 1024
    with pytest.raises(RuntimeError):
        F.fps(dev(torch.zeros(1, 20000, 3)), 8)                                                      # N <= 16384
    with pytest.raises(RuntimeError):
        F.attention(dev(torch.zeros(1, 600, 256)), dev(torch.zeros(1, 600, 256)), dev(torch.zeros(1, 600, 256)), 4)   # K/V of one head in 64 KB
    with pytest.raises(RuntimeError):
        F.linear(torch.zeros(4, 8), torch.zeros(8, 8))                                               # CPU tensors: no fallback
    # zero rows: nothing is launched, shapes are kept, gradients are zeros
    x = dev(torch.zeros(0, 64)).requires_grad_()
    w = dev(rnd(32, 64)).requires_grad_()
    y = F.linear(x, w)
    assert y.shape == (0, 32)
    y.sum().backward()
    assert w.grad.shape == w.shape and float(w.grad.abs().sum()) == 0.0
    # a batch whose hands are all invalid: every per-sample term is weighted by 0, the loss stays finite
    got = F.rowloss(dev(rnd(2, 3, 5)), dev(rnd(2, 3, 5, seed=1)), 2, 'l1') * dev(torch.zeros(2, 3))
    assert torch.isfinite(got).all() and float(got.abs().sum()) == 0.0


def test_stream_wait_orders_two_streams(F):
    """pdf_stream_wait(waiter, signaler): work issued on `waiter` afterwards sees everything `signaler` was given before --
    the fork / join the host layer uses around the weight-gradient side stream (functional.wgrad_stream / join_wgrad)."""
    import ctypes
    from pdfnet_amd import hip
    L = hip.lib()
    a, b = torch.cuda.Stream(), torch.cuda.Stream()
    ra, rb = ctypes.c_void_p(a.cuda_stream), ctypes.c_void_p(b.cuda_stream)
    assert L.pdf_stream_wait(ra, ra) == 0                                  # a stream never waits for itself
    x = dev(rnd(512, 2048))
    w = dev(rnd(2048, 2048, seed=1))
    torch.cuda.synchronize()
    for rep in range(2000 // 100):                                         # > the 1,024-slot event ring in total: slots are reused
        with torch.cuda.stream(a):
            y = x
            for _ in range(6):                                             # a dependent chain that keeps stream a busy
                y = F.linear(y, w) * 1e-1
        for _ in range(100):
            L.pdf_stream_wait(rb, ra)
        with torch.cuda.stream(b):
            z = y.sum()                                                    # would read a half-written y without the wait
        y.record_stream(b)
        L.pdf_stream_wait(ra, rb)
        torch.cuda.synchronize()
        assert float(z) == float(y.sum()), rep
    # the context manager built on it leaves the current stream as it found it, and its kernels land on the side stream
    cur = torch.cuda.current_stream().cuda_stream
    with F.wgrad_stream(True, x):
        side = torch.cuda.current_stream().cuda_stream
        assert side != cur or not F.ASYNC_WGRAD
    assert torch.cuda.current_stream().cuda_stream == cur
    F.join_wgrad()
    torch.cuda.synchronize()


def test_plain_c_client_of_the_c_abi(F, tmp_path):
    """The boundary without Python or torch: a C99 program (gcc) allocates with the HIP runtime, calls pdf_linear_fwd /
    pdf_stream_wait / pdf_linear_bwd_weight through include/pdfnet_hip.h and checks them against host loops."""
    import subprocess
    from tests.util import build_c_client
    r = subprocess.run([build_c_client(tmp_path)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "c_client: ok" in r.stdout, r.stdout


def test_gradient_chain_of_a_map_with_several_consumers(F):
    """max pool (skip) -> row gather (chain) -> row gather (chain) -> window gather (chain) on ONE feature map: the backward
    passes fill one gradient tensor; the result equals the sum autograd forms from independent consumers."""
    B, C, H, W = 2, 64, 32, 32
    x = rnd(B, C, H, W, seed=1)
    ind1 = torch.randint(0, H * W, (B, 40), generator=torch.Generator().manual_seed(2))
    ind2 = torch.randint(0, H * W, (B, 24), generator=torch.Generator().manual_seed(3))
    ind3 = torch.tensor([[0, H * W - 1], [W + 1, 5 * W + 7]])                 # windows that leave the map / overlap
    gp, g1, g2, g3 = rnd(B, C, 16, 16, seed=4), rnd(B, 40, C, seed=5), rnd(B, 24, C, seed=6), rnd(B * 2, C, 5, 5, seed=7)

    def run(chained):
        xd = dev(x).contiguous(memory_format=torch.channels_last).requires_grad_()
        if chained:
            p, a = F.maxpool3s2(xd, skip=True)
            r1, a = F.gather_rows(a, dev(ind1), chain=True)
            r2, a = F.gather_rows(a, dev(ind2), chain=True)
            w, a = F.window_gather(a, dev(ind3), 2, True)                      # the last alias stays unused
        else:
            p, r1, r2, w = F.maxpool3s2(xd), F.gather_rows(xd, dev(ind1)), F.gather_rows(xd, dev(ind2)), F.window_gather(xd, dev(ind3), 2)
        gw = dev(g3).contiguous(memory_format=torch.channels_last)
        torch.autograd.backward([p, r1, r2, w], [dev(gp).contiguous(memory_format=torch.channels_last), dev(g1), dev(g2), gw])
        return [t.detach().cpu() for t in (p, r1, r2, w)], xd.grad.detach().cpu()
    (oc, gc), (ou, gu) = run(True), run(False)
    for a, b in zip(oc, ou):
        assert torch.equal(a, b)
    close(gc, gu, 1e-5, rtol=1e-6, what="chained gradient of the shared map")
    # a chain whose first outputs are unused still delivers the later consumers' gradients
    xd = dev(x).contiguous(memory_format=torch.channels_last).requires_grad_()
    r1, a = F.gather_rows(xd, dev(ind1), chain=True)
    r2, a = F.gather_rows(a, dev(ind2), chain=True)
    r2.backward(dev(g2))
    xr = dev(x).contiguous(memory_format=torch.channels_last).requires_grad_()
    F.gather_rows(xr, dev(ind2)).backward(dev(g2))
    close(xd.grad, xr.grad, 1e-6, what="chain with an unused consumer")


# (N, Cin, H, W, Cout, k, stride, pad, act): one per statistics-capable launch -- the LDS-halo 128x128 tile, igemm_nt 128x128,
# 128x64 (narrow output), 64x64 with K-step 32 and 16, the one-wave 32x32 tile, ragged M / N, the per-element (non-FAST) form
# the last entry (few tiles under a long reduction) takes the split-K launch, which has none: the BatchNorm must fall back
@pytest.mark.parametrize("R,C,N", [(4096, 64, 128), (20480, 128, 256), (20480, 64, 64), (1000, 64, 64)])
def test_lazy_batchnorm_applied_by_the_consuming_linear(F, R, C, N):
    """F.batch_norm(..., relu=True, lazy=True) -> F.linear: the normalised rows are never written, the GEMM reads
    relu(x * scale + shift) while it stages its operand (forward, and the weight gradient through the 64x64 and the LDS-DMA
    kernels).  Same values in, so the forward is bit-identical to the materialised form and the gradients agree to rounding.
    (R = 1000 is not a multiple of 16: the request is ignored and the ordinary path runs.)"""
    x = rnd(R, C, seed=5) * 1.3 + 0.2
    w = rnd(N, C, seed=6) / C ** 0.5
    b = rnd(N, seed=7)
    g, be = torch.rand(C) + 0.5, rnd(C, seed=8) * 0.3
    dy = rnd(R, N, seed=9)
    out = {}
    for lazy in (True, False):
        xd, wd, bd = dev(x).requires_grad_(), dev(w).requires_grad_(), dev(b).requires_grad_()
        gd, bed = dev(g).requires_grad_(), dev(be).requires_grad_()
        rm, rv = dev(torch.zeros(C)), dev(torch.ones(C))
        z = F.batch_norm(xd, gd, bed, rm, rv, True, 0.1, 1e-5, relu=True, lazy=lazy)
        assert (getattr(z, '_pdf_lazy', None) is not None) == (lazy and R % 16 == 0)
        y = F.linear(z, wd, bd, F.ACT_NONE, stats=True)
        y.backward(dev(dy))
        F.join_wgrad()
        out[lazy] = [t.detach().cpu() for t in (y, rm, rv, xd.grad, wd.grad, bd.grad, gd.grad, bed.grad)]
    assert torch.equal(out[True][0], out[False][0]), "forward of the lazy form differs"
    assert torch.equal(out[True][1], out[False][1]) and torch.equal(out[True][2], out[False][2])
    for a, b_, name in zip(out[True][3:], out[False][3:], ('dx', 'dw', 'db', 'dgamma', 'dbeta')):
        close(a, b_, 2e-5 * float(b_.abs().max()) + 1e-7, what=name)


STAT_CONVS = [(10, 256, 64, 64, 256, 3, 1, 1, 0, True), (10, 128, 64, 64, 256, 3, 1, 1, 0, True), (40, 64, 64, 64, 64, 3, 1, 1, 0, True),
              (24, 128, 32, 32, 128, 3, 1, 1, 1, True), (8, 128, 32, 32, 128, 3, 1, 1, 1, False),      # (the second: 256 tiles, split over K)
              (40, 64, 64, 64, 256, 1, 1, 0, 0, True), (4, 48, 16, 16, 96, 3, 1, 1, 0, True),
              (2, 80, 4, 4, 64, 3, 1, 1, 0, True), (3, 48, 17, 15, 72, 3, 2, 1, 0, True), (2, 6, 9, 9, 20, 3, 1, 1, 1, True),
              (4, 3, 256, 256, 64, 7, 2, 3, 0, True),          # the ResNet stem's own kernel (stem7x7_fwd_kernel)
              (2, 512, 4, 4, 64, 3, 1, 1, 0, False)]


@pytest.mark.parametrize("cfg", STAT_CONVS)
def test_batchnorm_statistics_from_the_gemm_epilogue(F, cfg, monkeypatch):
    """conv -> training BatchNorm with the statistics taken out of the GEMM accumulators (IGemm::stat, per-row-block Chan
    partials, fp64 combination in pdf_bn_train_fwd): same output, saved statistics, running statistics and gradients as the
    BatchNorm's own statistics pass over the stored tensor, and both agree with a float64 evaluation.  (The subject is the direct
    kernels' epilogue: the Winograd path, which has none -- its BatchNorm runs its own statistics pass -- is switched off here.)"""
    monkeypatch.setattr(F, "WINOGRAD", False)
    N, Cin, H, W, Cout, k, st, pad, act, has_epilogue = cfg
    x = (rnd(N, Cin, H, W, seed=3) * 1.5 + 0.4)
    w = rnd(Cout, Cin, k, k, seed=4) / (Cin * k * k) ** 0.5
    g, b = torch.rand(Cout) + 0.5, rnd(Cout, seed=2)
    rm0, rv0 = rnd(Cout, seed=5), torch.rand(Cout) + 0.5
    res = {}
    for mode in ('epilogue', 'pass'):
        xd = dev(x).contiguous(memory_format=torch.channels_last).requires_grad_()
        wd = dev(w).contiguous(memory_format=torch.channels_last).requires_grad_()
        gd, bd = dev(g).requires_grad_(), dev(b).requires_grad_()
        rm, rv = dev(rm0.clone()), dev(rv0.clone())
        y = F.conv2d(xd, wd, None, st, pad, act, stats=(mode == 'epilogue'))
        if mode == 'epilogue':
            assert (F.tile_stats_of(y) is not None) == has_epilogue, "statistics epilogue expected: %s" % has_epilogue
        else:
            assert F.tile_stats_of(y) is None
        # (no ReLU here: the two paths' (scale, shift) differ in the last bit, which would flip the mask of the odd element whose
        # normalised value is within 1e-7 of zero -- an O(1) change of that element's gradient in either valid evaluation, about
        # once per 10^7 elements, i.e. a flaky element-wise comparison.  The mask logic itself is test_batchnorm's subject.)
        out = F.batch_norm(y, gd, bd, rm, rv, True, 0.1, 1e-5, False)
        out.backward(dev(rnd(*out.shape, seed=6)).contiguous(memory_format=torch.channels_last))
        F.join_wgrad()
        torch.cuda.synchronize()
        res[mode] = (out.detach(), rm, rv, xd.grad, wd.grad, gd.grad, bd.grad, y.detach())
    for a, c, what in zip(res['epilogue'], res['pass'], ('out', 'running_mean', 'running_var', 'dx', 'dw', 'dgamma', 'dbeta', 'y')):
        close(a, c.cpu(), 2e-5, rtol=2e-5, what=what)
    y64 = res['pass'][7].cpu().double().permute(0, 2, 3, 1).reshape(-1, Cout)
    mean, var = y64.mean(0), y64.var(0, unbiased=True)
    close(res['epilogue'][1], (0.9 * rm0.double() + 0.1 * mean).float(), 1e-5, rtol=1e-5, what="running_mean vs float64")
    close(res['epilogue'][2], (0.9 * rv0.double() + 0.1 * var).float(), 1e-5, rtol=2e-5, what="running_var vs float64")


def test_winograd_layers_equal_the_direct_kernels_through_a_batchnorm(F, monkeypatch):
    """A stride-1 3x3 convolution with >= 128 channels takes the Winograd path (csrc/winograd.hip: forward, backward-data, weight gradient);
    it has no statistics epilogue, so the BatchNorm behind it runs its own pass.  Output, running statistics and all gradients against the
    direct kernels (bars: F(4x4) carries ~1e-5 relative error, see tests/test_headline_gpu.py)."""
    N, Cin, H, W, Cout = 10, 256, 64, 64, 256
    x = (rnd(N, Cin, H, W, seed=3) * 1.5 + 0.4)
    w = rnd(Cout, Cin, 3, 3, seed=4) / (Cin * 9) ** 0.5
    bconv = rnd(Cout, seed=8)
    g, b = torch.rand(Cout) + 0.5, rnd(Cout, seed=2)
    res = {}
    for mode in (True, False):
        monkeypatch.setattr(F, "WINOGRAD", mode)
        xd = dev(x).contiguous(memory_format=torch.channels_last).requires_grad_()
        wd = dev(w).contiguous(memory_format=torch.channels_last).requires_grad_()
        cb = dev(bconv).requires_grad_()
        gd, bd = dev(g).requires_grad_(), dev(b).requires_grad_()
        rm, rv = dev(torch.zeros(Cout)), dev(torch.ones(Cout))
        y = F.conv2d(xd, wd, cb, 1, 1, 0, stats=True)
        if mode:
            assert F.tile_stats_of(y) is None                 # (the Winograd path reports no statistics: stats_tiles == 0)
        out = F.batch_norm(y, gd, bd, rm, rv, True, 0.1, 1e-5, False)
        out.backward(dev(rnd(*out.shape, seed=6)).contiguous(memory_format=torch.channels_last))
        F.join_wgrad()
        torch.cuda.synchronize()
        res[mode] = (out.detach(), rm, rv, xd.grad, wd.grad, cb.grad, gd.grad, bd.grad)
    for a, c, what in zip(res[True], res[False], ('out', 'running_mean', 'running_var', 'dx', 'dw', 'dbias', 'dgamma', 'dbeta')):
        if what == 'dbias':                                   # a bias in front of a training BatchNorm: exactly zero in exact arithmetic --
            for t in (a, c):                                  # both paths leave rounding noise of the sum over 40,960 pixels
                assert float(t.abs().max()) <= 1e-5 * float(res[False][4].abs().max()), what
            continue
        close(a, c.cpu(), 3e-4, rtol=5e-5, what=what)


def test_statistics_epilogue_survives_a_large_common_offset(F):
    """mean >> std: sums of x and x^2 would cancel catastrophically in fp32; the epilogue's shifted / Chan-combined partials do
    not (1x1 convolution with a large bias-like input channel: every output sits at ~1000 +- 1)."""
    N, Cin, H, W, Cout = 4, 64, 32, 32, 128
    x = rnd(N, Cin, H, W, seed=1)
    x[:, 0] = 1000.0
    w = rnd(Cout, Cin, 1, 1, seed=2) * 0.1
    w[:, 0] = 1.0
    xd, wd = dev(x).contiguous(memory_format=torch.channels_last), dev(w).contiguous(memory_format=torch.channels_last)
    y = F.conv2d(xd, wd, None, 1, 0, 0, stats=True)
    assert F.tile_stats_of(y) is not None
    rm, rv = dev(torch.zeros(Cout)), dev(torch.ones(Cout))
    out = F.batch_norm(y, dev(torch.ones(Cout)), dev(torch.zeros(Cout)), rm, rv, True, 1.0, 1e-5, False)
    y64 = y.cpu().double().permute(0, 2, 3, 1).reshape(-1, Cout)
    close(rm, y64.mean(0).float(), 1e-6, rtol=1e-6, what="mean")
    close(rv, y64.var(0, unbiased=True).float(), 1e-6, rtol=1e-4, what="var")
    ref = ((y64 - y64.mean(0)) / (y64.var(0, unbiased=False) + 1e-5).sqrt()).float()
    close(out.permute(0, 2, 3, 1).reshape(-1, Cout), ref, 2e-3, what="normalised")          # (fp32 spacing at 1000 is 6e-5: y itself carries it)


@pytest.mark.parametrize("shape", [(2, 128, 16, 16), (3, 128, 64, 64), (2, 256, 8, 8), (2, 64, 32, 32), (1, 128, 5, 7), (2, 16, 13, 14)])
def test_upsample2x_backward_at_the_decoders_shapes(F, shape):
    """pdf_upsample2x_bwd (gather form, one thread per input pixel and channel quad) at the channel counts of the dense decoders (128) and the other
    float4 cases, incl. odd sizes, against ATen's bilinear backward on the CPU.  (Round 6 tried an LDS-staged form of the gather -- a block stages the dy
    rows and columns of its input row segment once: bit-identical, but 2.3x slower, 64 KB of LDS per block leave two blocks per CU and nothing to hide the
    staging behind; profiles/r06_up2_lds.txt.)"""
    x = rnd(*shape, seed=sum(shape))
    xr = x.clone().requires_grad_()
    ref = TF.interpolate(xr, scale_factor=2, mode='bilinear', align_corners=True)
    gy = rnd(*ref.shape, seed=7)
    ref.backward(gy)
    xd = dev(x).requires_grad_()
    out = F.upsample2x(xd)
    out.backward(dev(gy))
    close(out, ref, 2e-6, what="upsample %s" % (shape,))
    close(xd.grad, xr.grad, 1e-5, what="upsample dx %s" % (shape,))
