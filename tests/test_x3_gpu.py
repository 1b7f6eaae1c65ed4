"""GPU: the public x3 entry points of the C ABI (include/pdfnet_hip.h "x3 arithmetic", csrc/gemm_x3.hip) on their own -- the split, the NT
product and the weight-gradient-shaped TN product -- against float64, with the native fp32-MFMA batched products (pdf_batched_gemm_nt / _tn)
computed on the same operands beside them.

The gate x3 ships under: its error against float64 is not larger than the native fp32 MFMA kernel's.
tools/x3_bench.py measures it on the step's shapes (profiles/r06_x3_bench.txt: rms 0.31-0.37x native); this file keeps it enforced, and
adds ragged shapes, every tile variant, and the argument checks."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def L():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from pdfnet_amd import hip
    return hip.lib()


def _ptr(t):
    from pdfnet_amd.hip import ptr
    return ptr(t)


def _stream():
    from pdfnet_amd.hip import stream
    return stream()


def _split3(L, x):
    o = torch.empty((3,) + tuple(x.shape), dtype=torch.bfloat16, device=x.device)
    L.pdf_x3_split(_ptr(x), _ptr(o), x.numel(), x.numel(), _stream())
    return o


def _spread(shape, seed, scale):
    """N(0, 1) times a per-plane factor over two decades (the spread of Winograd transform-domain values)."""
    g = torch.Generator(device="cuda").manual_seed(seed)
    x = torch.randn(shape, device="cuda", generator=g)
    f = torch.logspace(-1, 1, shape[0], device="cuda").view(-1, *([1] * (len(shape) - 1))) if shape[0] > 1 else 1.0
    return (x * f * scale).contiguous()


def _errs(c, ref):
    d = (c.double() - ref).abs()
    return d.max().item() / ref.abs().max().item(), (d.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item()


def _fp32_bar(K):
    """rms error (relative to the rms of the result) allowed to a K-term dot product computed in fp32."""
    return 3e-8 * K ** 0.5 + 5e-8


def test_split_is_exact(L):
    """a == h + m + l for every finite fp32 in the normal range: three bf16 significands (8 + 8 + 8 bits) hold the 24 bits of a."""
    g = torch.Generator(device="cuda").manual_seed(1)
    mant = (1.0 + torch.rand(1 << 20, device="cuda", generator=g)) * (torch.randint(0, 2, (1 << 20,), device="cuda", generator=g).float() * 2 - 1)
    expo = torch.randint(-100, 100, (1 << 20,), device="cuda", generator=g).float()
    x = mant * torch.exp2(expo)                                   # (|a| >= 2^-100: the last bit of a, 2^-123, is still a normal bf16 / fp32)
    x[:16] = 0.0
    x[16:32] = torch.tensor([1.0, -1.0, 3.0, 1.0 + 2.0 ** -23, 1.0 - 2.0 ** -24, 2.0 ** -100, -2.0 ** 100, 255.0, 256.0, 257.0, 65535.0, 65537.0,
                             0.1, -0.3, 1e-30, 1e30], device="cuda")
    s3 = _split3(L, x)
    back = (s3[0].float() + s3[1].float()) + s3[2].float()        # (h + m has <= 16 significant bits: every step here is exact in fp32)
    torch.cuda.synchronize()
    assert torch.equal(back, x), (back - x).abs().max().item()
    nz = x != 0
    assert float((s3[1].float().abs()[nz] / x.abs()[nz]).max()) <= 2.0 ** -8 and float((s3[2].float().abs()[nz] / x.abs()[nz]).max()) <= 2.0 ** -16


# batch, M, N, K: the step's shapes at a fraction of their plane / row count, then ragged ones (edge tiles in M and N, one K step, many K steps)
NT_SHAPES = [(2, 1024, 256, 1024, True), (2, 1024, 1024, 256, True), (1, 2048, 512, 2048, True), (2, 512, 256, 256, True),
             (3, 200, 72, 96, False), (1, 1000, 264, 32, False), (2, 257, 129, 160, False), (1, 33, 7, 64, False)]


@pytest.mark.parametrize("batch,M,N,K,gate", NT_SHAPES)
def test_nt_product_against_float64_beside_the_native_kernel(L, batch, M, N, K, gate):
    A = _spread((batch, M, K), 2, 1.0)
    B = _spread((batch, N, K), 3, 0.05)
    ref = torch.bmm(A.double(), B.double().transpose(1, 2))
    C = torch.empty(batch, M, N, device="cuda")
    L.pdf_batched_gemm_nt(_ptr(A), _ptr(B), _ptr(C), batch, M * K, N * K, M * N, M, N, K, _stream())
    e_nat = _errs(C, ref)
    A3, B3 = _split3(L, A), _split3(L, B)
    worst = (0.0, 0.0)
    for variant in (-1, 0, 1, 2, 3, 4, 5):
        C.fill_(float('nan'))
        L.pdf_x3_batched_gemm_nt(_ptr(A3), A.numel(), _ptr(B3), B.numel(), _ptr(C), batch, M * K, N * K, M * N, M, N, K, variant, 6, _stream())
        torch.cuda.synchronize()
        assert torch.isfinite(C).all(), "variant %d left elements unwritten" % variant
        e = _errs(C, ref)
        worst = (max(worst[0], e[0]), max(worst[1], e[1]))
        print("  variant %d: max %.3e rms %.3e (native %.3e %.3e)" % (variant, e[0], e[1], e_nat[0], e_nat[1]))
        # fp32-grade on every shape and tile: the rms rounding error of a K-term fp32 dot product grows like sqrt(K) (measured 0.9e-7 at K = 64,
        # 4.9e-7 at 1,024, 7.0e-7 at 2,048 -- the native kernel 1.3e-7 / 5.8e-7 / 8.2e-7)
        assert e[1] <= _fp32_bar(K) and e[0] <= 10 * _fp32_bar(K), (variant, e)
        if gate:                                            # the shipped shapes: not worse than the native fp32 MFMA kernel on the same operands
            assert e[1] <= 1.0 * e_nat[1] and e[0] <= 1.25 * e_nat[0], (variant, e, e_nat)
    print("NT %dx%dx%dx%d: native max %.2e rms %.2e; x3 (worst of 7 tile choices) max %.2e rms %.2e = %.2fx" %
          (batch, M, N, K, e_nat[0], e_nat[1], worst[0], worst[1], worst[1] / e_nat[1]))


def test_nt_product_count_is_what_sets_the_error(L):
    """nprod = 9 (all nine component products) changes nothing measurable; nprod = 3 (hh', hm', mh' only: ~2^-16) is visibly coarser -- the
    argument acts, and six is the smallest count that reaches fp32."""
    batch, M, N, K = 1, 512, 256, 512
    A, B = _spread((batch, M, K), 4, 1.0), _spread((batch, N, K), 5, 0.05)
    ref = torch.bmm(A.double(), B.double().transpose(1, 2))
    A3, B3 = _split3(L, A), _split3(L, B)
    C = torch.empty(batch, M, N, device="cuda")
    e = {}
    for n in (3, 6, 9):
        for variant in (3, 0):                               # (the 64x64 tile with three products is the one schedule with all fragment reads in slot 0)
            L.pdf_x3_batched_gemm_nt(_ptr(A3), A.numel(), _ptr(B3), B.numel(), _ptr(C), batch, M * K, N * K, M * N, M, N, K, variant, n, _stream())
            e[n, variant] = _errs(C, ref)[1]
        assert e[n, 3] == e[n, 0], e                         # same products, same order within an accumulator: the tile shape does not change a bit
        e[n] = e[n, 0]
    print("rms error vs float64 by product count: %s" % e)
    assert e[6] <= _fp32_bar(K) and e[9] <= 1.1 * e[6] and e[6] <= 1.1 * e[9]
    assert 10 * e[6] < e[3] < 1e-4


# batch, M (reduction rows), NI, NJ, splits
TN_SHAPES = [(2, 2048, 256, 1024, 4, True), (2, 2048, 256, 256, 4, True), (2, 2048, 128, 128, 8, True),
             (2, 512, 136, 72, 3, False), (1, 96, 8, 264, 1, False), (3, 160, 72, 40, 5, False)]


@pytest.mark.parametrize("batch,M,NI,NJ,splits,gate", TN_SHAPES)
def test_tn_product_against_float64_beside_the_native_kernel(L, batch, M, NI, NJ, splits, gate):
    P = _spread((batch, M, NI), 6, 0.05)
    Q = _spread((batch, M, NJ), 7, 1.0)
    ref = torch.bmm(P.double().transpose(1, 2), Q.double())

    def used(q):                                             # rows per split rounded up to q, as the library plans it
        rps = -(-(-(-M // splits)) // q) * q
        return -(-M // rps)
    slab = torch.zeros(batch, splits, NI, NJ, device="cuda")
    L.pdf_batched_gemm_tn(_ptr(P), _ptr(Q), _ptr(slab), batch, M * NI, M * NJ, M, NI, NJ, splits, _stream())
    un = used(16)
    e_nat = _errs(slab.flatten()[:batch * un * NI * NJ].view(batch, un, NI, NJ).sum(1), ref)
    P3, Q3 = _split3(L, P), _split3(L, Q)
    ux = used(32)
    for variant in (-1, 0, 1, 2, 3):
        slab.fill_(float('nan'))
        L.pdf_x3_batched_gemm_tn(_ptr(P3), P.numel(), _ptr(Q3), Q.numel(), _ptr(slab), batch, M * NI, M * NJ, M, NI, NJ, splits, variant, 6, _stream())
        torch.cuda.synchronize()
        c = slab.flatten()[:batch * ux * NI * NJ].view(batch, ux, NI, NJ)
        assert torch.isfinite(c).all(), "variant %d left slab elements unwritten" % variant
        e = _errs(c.sum(1), ref)
        print("  variant %d: max %.3e rms %.3e (native %.3e %.3e)" % (variant, e[0], e[1], e_nat[0], e_nat[1]))
        assert e[1] <= _fp32_bar(M) and e[0] <= 10 * _fp32_bar(M), (variant, e)
        if gate:
            assert e[1] <= 1.0 * e_nat[1] and e[0] <= 1.25 * e_nat[0], (variant, e, e_nat)
    print("TN %dx%dx%dx%d / %d: native max %.2e rms %.2e; x3 max %.2e rms %.2e" % (batch, M, NI, NJ, splits, e_nat[0], e_nat[1], e[0], e[1]))


def test_argument_checks(L):
    """A shape the kernels do not take is refused (PDF_E_BADARG -> RuntimeError through the binding), never run."""
    x = torch.zeros(3, 64, 48, dtype=torch.bfloat16, device="cuda")
    c = torch.zeros(64, 64, device="cuda")
    with pytest.raises(RuntimeError):                        # K % 32 != 0
        L.pdf_x3_batched_gemm_nt(_ptr(x), 64 * 48, _ptr(x), 64 * 48, _ptr(c), 1, 0, 0, 0, 64, 64, 48, 0, 6, _stream())
    with pytest.raises(RuntimeError):                        # reduction rows % 32 != 0
        L.pdf_x3_batched_gemm_tn(_ptr(x), 64 * 48, _ptr(x), 64 * 48, _ptr(c), 1, 0, 0, 48, 64, 64, 1, 0, 6, _stream())
    f = torch.zeros(20, device="cuda")
    with pytest.raises(RuntimeError):                        # n % 8 != 0
        L.pdf_x3_split(_ptr(f), _ptr(x), 20, 24, _stream())
