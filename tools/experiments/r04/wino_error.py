"""r04: the arithmetic error of Winograd F(2x2,3x3) and F(4x4,3x3) in float32, independent of any kernel: the algorithms are emulated with
PyTorch on the CPU in float32 and compared with a float64 convolution (and with the same emulation in float64, which shows the algebra is
exact).  Printed: max |error| on outputs whose largest value is ~4.4, next to a direct float32 convolution's."""
import torch
torch.manual_seed(0)


def wino(x, w, m):
    if m == 2:
        BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=x.dtype)
        G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=x.dtype)
        AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=x.dtype)
    else:
        BT = torch.tensor([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]], dtype=x.dtype)
        G = torch.tensor([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]], dtype=x.dtype)
        AT = torch.tensor([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], dtype=x.dtype)
    a = m + 2
    N, C, H, W = x.shape
    O = w.shape[0]
    tiles = torch.nn.functional.pad(x, (1, 1, 1, 1)).unfold(2, a, m).unfold(3, a, m)
    V = torch.einsum('ij,ncthjk,lk->ncthil', BT, tiles, BT)
    U = torch.einsum('ij,ocjk,lk->ocil', G, w, G)
    M = torch.einsum('ncthil,ocil->nothil', V, U)
    Y = torch.einsum('ij,nothjk,lk->nothil', AT, M, AT)
    return Y.permute(0, 1, 2, 4, 3, 5).reshape(N, O, H, W)


for C in (256, 1024):
    x = torch.randn(1, C, 24, 24)
    w = torch.randn(64, C, 3, 3) * (C * 9) ** -0.5
    ref = torch.nn.functional.conv2d(x.double(), w.double(), padding=1)
    d32 = torch.nn.functional.conv2d(x, w, padding=1)
    for m in (2, 4):
        print("C=%d F(%dx%d,3x3): float32 error %.2e (the float64 emulation: %.1e); direct float32 convolution %.2e; max|ref| %.2f" % (
            C, m, m, float((wino(x, w, m).double() - ref).abs().max()), float((wino(x.double(), w.double(), m) - ref).abs().max()),
            float((d32.double() - ref).abs().max()), float(ref.abs().max())))
