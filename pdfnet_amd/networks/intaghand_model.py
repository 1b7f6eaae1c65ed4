"""MI355X-native counterpart of lib/models/networks/intaghand_model.py: `HandNET_GCN` and the factory
`load_model_intag(opt)` with the reference's forward signature, return structure and state_dict keys, so
it drops into `ModleWithLoss` (lib/trains/base_trainer.py:24-78) / demo.py:202 unchanged.
"""
import torch
import torch.nn as nn

from .. import functional as F
from .intaghand_decoder import load_decoder
from .intaghand_encoder import load_encoder
from .layers import BatchNorm


class HandNET_GCN(nn.Module):
    def __init__(self, encoder, mid_model, decoder, run_mid_model=True):
        super().__init__()
        self.encoder = encoder
        self.mid_model = mid_model
        self.decoder = decoder
        # mid_model's fmaps are dead downstream but its BatchNorm running statistics are
        # state the reference updates every step; keep it on by default for state parity.
        self.run_mid_model = run_mid_model
        self.defer_mid_model = False       # set by ModleWithLoss: mid_model on a side stream, joined after the loss
        self.defer_lossless_heads = False  # set by the Trainer (CtdetLoss only): wh / params heads issued after the loss, see run_deferred_heads
        self._deferred = []
        self._lazy_heads = None

    def forward(self, img, choose, cloud, depth, ind, K_new, valid):
        # a previous train-mode call that nobody joined (a direct `model(...)` outside ModleWithLoss): its mid_model pass still
        # owns a fork stream and writes BatchNorm running statistics this call is about to read / update
        self.join_deferred()
        st = self.encoder.trunk(img, ind, choose, cloud, depth, K_new, valid)
        img_fmaps, ind = st['img_fmaps'], st['ind']
        # The dense branches are issued BEFORE the launch-bound mesh decoder.  Issuing them after it (so that their heavy
        # convolutions overlap the decoder's small kernels in forward and backward) was measured: 267-272 vs 314 img/s in
        # one session -- the big kernels starve the critical decoder -> PointNet++ -> trunk backward chain.  Round 2 re-measured it
        # with the branches waiting on a post-trunk event only: 388 vs 407 img/s; with the decoder on a high-priority stream: 283.
        gl, gr = img_fmaps[0][:, 0], img_fmaps[0][:, 1]                 # what mid_model hands on (intaghand_encoder.py:881)
        # ... and joined only AFTER the mesh decoder has been issued: in the forward the decoder's small kernels run next to
        # the branches' convolutions (404.5 -> 406.8 img/s), while the autograd order of the backward stays as it was.
        lazy = self.defer_lossless_heads and self.training and torch.is_grad_enabled()
        join_dense = self.encoder.dense_branches(st, defer=True, lazy_heads=lazy)
        result, paramsDict, handDictList, otherInfo = self.decoder(gl, gr)
        hms, mask, ret, hms_fmaps, dp_fmaps = join_dense()
        if self.run_mid_model:
            def run_mid():
                with torch.no_grad():
                    if self.defer_mid_model and self.training:
                        # the pass may still run while the backward releases these tensors: the allocator must not hand their memory out before it ends
                        for t in list(img_fmaps) + list(hms_fmaps) + list(dp_fmaps):
                            if torch.is_tensor(t) and t.is_cuda:
                                t.record_stream(torch.cuda.current_stream())
                    self.mid_model(img_fmaps, hms_fmaps, dp_fmaps)      # live state: its BN running statistics only
            if self.defer_mid_model and self.training:
                # nothing reads its outputs: 0.9 ms of convolutions that need not sit between the mesh decoder and the loss.
                # The caller that set `defer_mid_model` calls `join_deferred()` once the loss has been issued.
                self._deferred.append(F.fork(run_mid))
            else:
                run_mid()
        BatchNorm.flush_counters()
        self._lazy_heads = (ret.pop('_lazy_heads'), ret) if '_lazy_heads' in ret else None
        otherInfo['hms'] = hms
        otherInfo['mask'] = mask
        otherInfo['ret'] = ret
        otherInfo['ind'] = ind
        otherInfo['converter_left'] = self.decoder.converter['left']
        otherInfo['converter_right'] = self.decoder.converter['right']
        return result, paramsDict, handDictList, otherInfo

    def run_deferred_heads(self):
        """Issue the wh / params heads the last forward left out (`defer_lossless_heads`) on a side stream and fill them into the `ret` dict that
        forward returned; `join_deferred` makes them visible to the current stream."""
        if getattr(self, '_lazy_heads', None) is None:
            return
        fn, ret = self._lazy_heads
        self._lazy_heads = None
        f = F.fork(fn)
        ret.update(f.out)
        self._deferred.append(f)

    def join_deferred(self):
        """Make the current stream wait for the work `forward` left on side streams (see `defer_mid_model`, `defer_lossless_heads`)."""
        self.run_deferred_heads()                              # (nobody issued them: do it now)
        for f in self._deferred:
            f.join()
        self._deferred.clear()

    # readers of the module's state first wait for the deferred mid_model pass (it writes BatchNorm running statistics)
    def state_dict(self, *args, **kwargs):
        self.join_deferred()
        return super().state_dict(*args, **kwargs)

    def train(self, mode=True):
        self.join_deferred()
        return super().train(mode)


def load_model_intag(opt):
    """Reads the same `opt` fields as the reference factory (intaghand_model.py:49-67)."""
    encoder, mid_model = load_encoder(opt)
    decoder = load_decoder(opt, mid_model.get_info())
    return HandNET_GCN(encoder, mid_model, decoder)
