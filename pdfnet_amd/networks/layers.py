"""Parameter-holding building blocks with the reference's state_dict key names whose forward runs on
the HIP kernels (pdfnet_amd.functional).  Initialisers follow the reference's defaults
 so random-init benchmarks behave like the reference's from-scratch runs.
"""
import math

import torch
import torch.nn as nn

from .. import functional as F


class Conv2d(nn.Module):
    """nn.Conv2d counterpart (weight logical OIHW, stored channels_last = [Cout][KH][KW][Cin])."""

    def __init__(self, cin, cout, k, stride=1, pad=0, bias=True):
        super().__init__()
        self.stride, self.pad = stride, pad
        w = torch.empty(cout, cin, k, k)
        nn.init.kaiming_uniform_(w, a=math.sqrt(5))
        self.weight = nn.Parameter(w.contiguous(memory_format=torch.channels_last))
        if bias:
            bound = 1.0 / math.sqrt(cin * k * k)
            self.bias = nn.Parameter(torch.empty(cout).uniform_(-bound, bound))
        else:
            self.register_parameter('bias', None)

    def forward(self, x, act=F.ACT_NONE, stats=False):
        """stats: the output goes straight into a training-mode BatchNorm (F.conv2d)."""
        return F.conv2d(x, self.weight, self.bias, self.stride, self.pad, act, stats)


class ConvTranspose2d(nn.Module):
    """nn.ConvTranspose2d counterpart (weight logical [Cin,Cout,KH,KW], stored channels_last)."""

    def __init__(self, cin, cout, k, stride, pad):
        super().__init__()
        self.stride, self.pad = stride, pad
        w = torch.empty(cin, cout, k, k)
        nn.init.kaiming_uniform_(w, a=math.sqrt(5))
        self.weight = nn.Parameter(w.contiguous(memory_format=torch.channels_last))
        bound = 1.0 / math.sqrt(cout * k * k)
        self.bias = nn.Parameter(torch.empty(cout).uniform_(-bound, bound))

    def forward(self, x):
        return F.deconv2d(x, self.weight, self.bias, self.stride, self.pad)


class Linear(nn.Module):
    def __init__(self, cin, cout, bias=True):
        super().__init__()
        w = torch.empty(cout, cin)
        nn.init.kaiming_uniform_(w, a=math.sqrt(5))
        self.weight = nn.Parameter(w)
        if bias:
            bound = 1.0 / math.sqrt(cin)
            self.bias = nn.Parameter(torch.empty(cout).uniform_(-bound, bound))
        else:
            self.register_parameter('bias', None)

    def forward(self, x, act=F.ACT_NONE):
        return F.linear(x, self.weight, self.bias, act)


class PointConv(nn.Module):
    """1x1 nn.Conv2d applied to point rows: weight keeps the reference shape [Cout,Cin,1,1]; the input
    rows may be zero-padded to `kpad` channels (16-float aligned rows for the fast GEMM path)."""

    def __init__(self, cin, cout):
        super().__init__()
        w = torch.empty(cout, cin, 1, 1)
        nn.init.kaiming_uniform_(w, a=math.sqrt(5))
        self.weight = nn.Parameter(w)
        bound = 1.0 / math.sqrt(cin)
        self.bias = nn.Parameter(torch.empty(cout).uniform_(-bound, bound))

    def matrix(self, kpad, npad=None):
        """[npad or Cout, kpad] weight matrix (zero columns for the padded input channels, zero rows for padded outputs): one launch
        (F.pad2d) instead of a fill and a strided copy per padded side."""
        w = self.weight.flatten(1)
        return F.pad2d(w, npad if npad is not None else w.shape[0], kpad)

    def forward(self, x, act=F.ACT_NONE, npad=None, stats=False):
        """npad: also pad the OUTPUT channels (zero weight rows, zero bias) -- a 131- / 259-wide output has rows that are not
        16-byte aligned, which sends its backward-data and weight-gradient GEMMs down the per-element path.
        stats: the output goes straight into a training-mode BatchNorm (F.linear)."""
        w, b = self.matrix(x.shape[-1], npad), self.bias
        if npad is not None and npad != b.shape[0]:
            b = F.pad2d(b.view(1, -1), 1, npad).view(-1)
        return F.linear(x, w, b, act, stats=stats)


class BatchNorm(nn.Module):
    """nn.BatchNorm2d / BatchNorm1d counterpart over rows (per-channel statistics)."""

    def __init__(self, c, momentum=0.1, eps=1e-5):
        super().__init__()
        self.momentum, self.eps = momentum, eps
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))
        self.register_buffer('running_mean', torch.zeros(c))
        self.register_buffer('running_var', torch.ones(c))
        self.register_buffer('num_batches_tracked', torch.tensor(0, dtype=torch.long))
        self._pending = 0

    # `num_batches_tracked += 1` is one launch per layer call (84 per step); the increments are counted on the host and
    # applied by one multi-tensor add (flush_counters: end of HandNET_GCN.forward, and before any state_dict read).
    _dirty = []

    @staticmethod
    def flush_counters():
        mods, BatchNorm._dirty = BatchNorm._dirty, []
        if mods:
            torch._foreach_add_([m.num_batches_tracked for m in mods], [m._pending for m in mods])
            for m in mods:
                m._pending = 0

    def _save_to_state_dict(self, destination, prefix, keep_vars):
        BatchNorm.flush_counters()
        super()._save_to_state_dict(destination, prefix, keep_vars)

    def forward(self, x, relu=False, res=None, lazy=False):
        if self.training:
            if self._pending == 0:
                BatchNorm._dirty.append(self)
            self._pending += 1
        return F.batch_norm(x, self.weight, self.bias, self.running_mean, self.running_var,
                            self.training, self.momentum, self.eps, relu, res, lazy)


    def relu_max_over_k(self, x, K):
        """relu(self(x)) followed by the max over groups of K consecutive rows, fused (F.bn_relu_max_over_k)."""
        if self.training:
            if self._pending == 0:
                BatchNorm._dirty.append(self)
            self._pending += 1
        return F.bn_relu_max_over_k(x, self.weight, self.bias, self.running_mean, self.running_var, K, self.training, self.momentum, self.eps)


class LayerNorm(nn.Module):
    def __init__(self, d, eps=1e-6):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(d))
        self.bias = nn.Parameter(torch.zeros(d))

    def forward(self, x):
        return F.layer_norm(x, self.weight, self.bias, self.eps)


class Embedding(nn.Module):
    def __init__(self, n, d):
        super().__init__()
        self.weight = nn.Parameter(torch.randn(n, d))


class Slot(nn.Module):
    """Parameter-free placeholder so nn.Sequential indices match the reference's (ReLU, pool, upsample)."""

    def forward(self, x):
        return x


def xavier_(module):
    """model_attn/*.py `weights_init`: xavier-uniform Linear/Conv weights, zero bias."""
    for m in module.modules():
        if isinstance(m, (Linear, Conv2d)):
            nn.init.xavier_uniform_(m.weight.data)
            if m.bias is not None:
                nn.init.constant_(m.bias.data, 0.0)


def kaiming_normal_(module):
    """intaghand_encoder.py:167-175 `weights_init` used on hms_decoder / dp_decoder."""
    for m in module.modules():
        if isinstance(m, (Linear, Conv2d)):
            nn.init.kaiming_normal_(m.weight.data)
            if isinstance(m, Linear) and m.bias is not None:
                nn.init.constant_(m.bias.data, 0.0)


def small_normal_(module, std=0.001):
    """intaghand_encoder.py:336-347 `fill_fc_weights`."""
    for m in module.modules():
        if isinstance(m, (Linear, Conv2d, ConvTranspose2d)):
            nn.init.normal_(m.weight.data, std=std)
            if m.bias is not None:
                nn.init.constant_(m.bias.data, 0.0)
