"""What does one more dependent launch cost on a stream?  Sequences of a large streaming kernel (x *= 1 over 256 MB) with and
without a tiny kernel (one block) between consecutive ones, and a chain of tiny kernels alone."""
import torch, time
dev = torch.device('cuda')
x = torch.ones(64 * 1024 * 1024, device=dev)
y = torch.ones(64 * 1024 * 1024, device=dev)
s = torch.zeros(64, device=dev)


def timed(fn, n):
    for _ in range(3):
        fn(10)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    fn(n)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


def big(n):
    for _ in range(n):
        x.mul_(1.0)


def big_tiny(n):
    for _ in range(n):
        x.mul_(1.0)
        s.add_(1.0)


def big2(n):
    for _ in range(n):
        x.mul_(1.0)
        y.mul_(1.0)


def big2_tiny(n):
    for _ in range(n):
        x.mul_(1.0)
        s.add_(1.0)
        y.mul_(1.0)
        s.add_(1.0)


def tiny(n):
    for _ in range(n):
        s.add_(1.0)


N = 300
a, b, c, d, t = timed(big, N), timed(big_tiny, N), timed(big2, N), timed(big2_tiny, N), timed(tiny, 2000)
print("big kernel alone            %7.1f us per iteration" % a)
print("big + tiny                  %7.1f us per iteration  -> the tiny launch costs %.1f us" % (b, b - a))
print("big(x) + big(y)             %7.1f us per iteration" % c)
print("big(x) tiny big(y) tiny     %7.1f us per iteration  -> %.1f us per tiny launch" % (d, (d - c) / 2))
print("chain of tiny kernels       %7.1f us per launch" % t)
