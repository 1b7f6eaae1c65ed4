"""Counterpart of the reference's live loss branch, `CtdetLoss.forward` (lib/trains/simplified.py:364-655),
for the H2O RGB-D two-hand task (flags of scripts/train.sh: --reproj_loss --bone_loss, dataset H2O).

SURVEY.md 8(f) row 1: the reference's loss is ~40 launch-bound element-wise / tiny-matmul ops per term and hand.  Here
every term is a HIP kernel of csrc/loss.hip over BOTH hands: `F.rowloss` (all L1 / MSE mesh terms), `F.face_loss` (normal +
edge-length), `F.dense_loss` (SmoothL1 mask + MSE heat-maps + focal centre loss, one forward and one backward launch),
`F.regress_joints_pair`; only index bookkeeping (stack / gather of ground truth, the pinhole projection of 42 joints and
the 20-bone direction term on [2,B,21,2]) is left to aten.  The step is sync-free, so it is hipGraph-capturable.
Differences from the reference are host-side only: no `.cpu()` debug dumps (simplified.py:527-596) and the
focal loss' `if num_pos.sum() == 0` host branch (lib/models/losses.py:161) is a device-side select.
"""
import os

import torch
import torch.nn as nn

from .. import functional as F

_BONES = [(0, 1), (1, 2), (2, 3), (3, 4), (0, 5), (5, 6), (6, 7), (7, 8), (0, 9), (9, 10), (10, 11), (11, 12),
          (0, 13), (13, 14), (14, 15), (15, 16), (0, 17), (17, 18), (18, 19), (19, 20)]      # losses.py:34-53


def sigmoid_clamped(x):
    """lib/models/utils.py:8-10 `_sigmoid` (out-of-place: the caller's logits stay intact)."""
    return torch.clamp(torch.sigmoid(x), min=1e-4, max=1 - 1e-4)


def bone_direction_loss(j2d, gt2d, a, c):
    """lib/models/losses.py:26-94 `get_bone_loss` with unit confidences -> [B]; a, c = bone end-point indices."""

    def unit(j):
        v = j.index_select(1, c) - j.index_select(1, a)                  # [B,20,2]
        return v / torch.sqrt((v ** 2).sum(-1, keepdim=True) + 1e-4)
    return ((unit(j2d) - unit(gt2d)) ** 2).sum(-1).mean(dim=1)


def perspective(pts, K):
    """Mano_render.py:203-209 `get_Landmarks_new`."""
    p = F.project_points(pts, K)                                          # pts [..., B, n, 3] @ K[b]^T
    return p[..., :2] / (p[..., 2:] + 1e-7)


def uv_root_3d(index, root_xy, root_z, K, input_res, down_ratio=4):
    """Mano_render.py:211-223 `get_uv_root_3d`."""
    g = input_res // down_ratio                                         # index [..., B], root_xy [..., B, 2], root_z [..., B]
    cx = (index % g) * down_ratio
    cy = (index // g) * down_ratio
    x = root_z * (root_xy[..., 0] + cx - K[:, 0, 2]) / (K[:, 0, 0] + 1e-7)
    y = root_z * (root_xy[..., 1] + cy - K[:, 1, 2]) / (K[:, 1, 1] + 1e-7)
    return torch.stack((x, y, root_z), -1).unsqueeze(-2)


def _pool4(x):
    """simplified.py:117-122 `mesh_downsample` applied twice (pair averages of pair averages along the vertex axis)."""
    *lead, N, Fd = x.shape
    return x.reshape(*lead, N // 2, 2, Fd).mean(-2).reshape(*lead, N // 4, 2, Fd).mean(-2)


def _pair(d):
    """{'left': t, 'right': t} -> [2, ...]; no copy when the two are the halves of one stacked tensor (our decoder's outputs)."""
    l, r = d['left'], d['right']
    b = l._base
    if (b is not None and b is r._base and b.dim() == l.dim() + 1 and b.shape[0] == 2 and b.shape[1:] == l.shape and b.is_contiguous()
            and l.is_contiguous() and r.is_contiguous() and l.data_ptr() == b.data_ptr()
            and r.data_ptr() == b.data_ptr() + l.numel() * l.element_size()):
        return b
    return torch.stack((l, r))


def projection_batch(scale, trans2d, pts, img_size):
    s = (scale * img_size).view(-1, 1, 1)
    t = (trans2d * img_size / 2 + img_size / 2).unsqueeze(1)
    return s * pts[..., :2] + t


FORK_DENSE_LOSS = os.environ.get("PDFNET_FORK_DENSE_LOSS", "1") != "0"


class CtdetLoss(nn.Module):
    """opt fields used: size_train, center_weight (200), reproj_weight (1), bone_dir_weight (200), down_ratio (4).
    `consts`: {'full_regressor_left/right' [21,778], 'faces_left/right' [1538,3] int64}."""

    def __init__(self, opt, consts):
        super().__init__()
        self.opt = opt
        for k, v in consts.items():
            self.register_buffer(k, v.clone(), persistent=False)
        self.register_buffer('faces_pair', torch.stack((consts['faces_left'], consts['faces_right'])).long(), persistent=False)
        self.register_buffer('bone_a', torch.tensor([b[0] for b in _BONES]), persistent=False)
        self.register_buffer('bone_c', torch.tensor([b[1] for b in _BONES]), persistent=False)
        self.register_buffer('_hand_scales', torch.tensor([1.0, 1.0, 1000.0, 1000.0, 1000.0, 1.0, 1.0]).view(7, 1, 1), persistent=False)
        self._coef_cache = {}

    def forward(self, result, paramsDict, handDictList, otherInfo, batch, mode, epoch):
        test = mode in ('val', 'test')
        if test:
            from ..networks.intaghand_encoder import nms_top1_centers
            ind = nms_top1_centers(sigmoid_clamped(otherInfo['ret']['hm']))                # :376-389
        else:
            ind = batch['ind']
        # dense-map terms on a forked stream (round 6): two launches forward and one backward that neither feed nor need the mesh terms -- they and
        # their backward (which hands its gradients to the dense decoders' own streams) leave the main chain (PDFNET_FORK_DENSE_LOSS=0: in line, last)
        fd = F.fork(lambda: self.dense_terms(otherInfo, batch)) if (FORK_DENSE_LOSS and not test) else None
        t = self.mesh_terms(result, paramsDict, handDictList, otherInfo['converter_left'], otherInfo['converter_right'],
                            batch, ind, test, epoch)
        if test:
            return t
        # (in line: created last, so that their backward is issued first, ahead of the launch-bound mesh terms)
        t.update(fd.join() if fd is not None else self.dense_terms(otherInfo, batch))
        mp = t.pop('_mesh_part', None)
        if mp is not None:                                   # fused mesh terms: their weighted sum came out of the kernel
            loss = mp + self.dense_part(t)
            stats = {k: t[k] for k in self._ORDER}
            stats['loss'] = loss
            return loss, stats, None, None
        return self.total(t, epoch)

    def dense_terms(self, otherInfo, batch):
        """The terms on the encoder's dense maps: hand masks (:368), joint heat-maps (:374), centre heat-map (:376,391) --
        SmoothL1, MSE and the focal loss on the clamped sigmoid, all in `F.dense_loss` (csrc/loss.hip)."""
        mask_loss, hms_loss, hm_loss = F.dense_loss(otherInfo['mask'], batch['mask'], otherInfo['hms'], batch['hms'],
                                                    otherInfo['ret']['hm'], batch['hm'])
        return {'mask_loss': mask_loss, 'hms_loss': hms_loss, 'hm_loss': hm_loss}

    def dense_part(self, t):
        """What dense_terms contribute to the total ([B]); total == dense_part + mesh_part up to summation order."""
        o = self.opt
        return getattr(o, 'center_weight', 200.0) * t['hm_loss'] + getattr(o, 'reproj_weight', 1.0) * (t['mask_loss'] * 2000 + t['hms_loss'] * 2000)

    def mesh_part(self, t, epoch):
        o = self.opt
        alpha = 0 if epoch < 20 else 1
        w = getattr(o, 'reproj_weight', 1.0)
        return (w * (t['root_loss'] + t['verts_loss'] * 500 + t['abs_verts_loss'] * 0.1 + t['verts2d_loss'] * 50 + t['norm_loss'] * 10 +
                     t['edge_loss'] * 2000 * alpha + t['gcn_loss'] * 100 + t['gcn_2d_loss'] * 50 + t['abs_joints_loss'] * 0.1 +
                     t['joints2d_loss'] * 1000 * alpha + t['joints_loss'] * 500) + getattr(o, 'bone_dir_weight', 200.0) * t['bone_direc_loss'])

    _ORDER = ('hm_loss', 'root_loss', 'verts_loss', 'abs_verts_loss', 'verts2d_loss', 'norm_loss', 'edge_loss', 'gcn_loss',
              'gcn_2d_loss', 'mask_loss', 'abs_joints_loss', 'hms_loss', 'joints2d_loss', 'joints_loss', 'bone_direc_loss')

    def coefficients(self, epoch):
        """The weights of the reference's sum (:610-640), in _ORDER."""
        o = self.opt
        alpha = 0 if epoch < 20 else 1                                                                      # :610
        w = getattr(o, 'reproj_weight', 1.0)
        return (getattr(o, 'center_weight', 200.0), w, w * 500, w * 0.1, w * 50, w * 10, w * 2000 * alpha, w * 100, w * 50,
                w * 2000, w * 0.1, w * 2000, w * 1000 * alpha, w * 500, getattr(o, 'bone_dir_weight', 200.0))

    def total(self, t, epoch):
        """The reference's weighted sum (:610-640) -> (loss [B], stats, None, None): one stacked product-sum over the 15 terms
        (the term-by-term expression was ~30 launches forward and ~45 backward on [B]-sized tensors)."""
        coefs = self.coefficients(epoch)
        terms = [t[k] for k in self._ORDER]
        dev = terms[0].device
        c = self._coef_cache.get((coefs, dev))
        if c is None:
            c = self._coef_cache[(coefs, dev)] = torch.tensor(coefs, dtype=torch.float32, device=dev).view(-1, 1)
        B = max(x.numel() for x in terms)
        loss = (torch.stack([x.expand(B) if x.dim() == 0 else x for x in terms]) * c).sum(0)
        stats = {k: t[k] for k in self._ORDER}
        stats['loss'] = loss
        return loss, stats, None, None

    def mesh_terms(self, result, paramsDict, handDictList, cl, cr, batch, ind, test, epoch):
        """Every term on the mesh decoder's outputs (:425-525).  cl / cr: the decoder's GCN <-> MANO vertex-order converters.
        Returns the dict of terms, or the evaluation tuple (:652-653) when `test`.

        Both hands are processed as one stacked [2, B, ...] tensor (row 0 left, row 1 right); each L1 / MSE term is one
        `F.rowloss` launch, normal + edge-length terms one `F.face_loss` launch.  nrm(x) = x/S*2 - 1 (:421) only rescales the
        MSE terms: mse(nrm(a), nrm(b)) = (2/S)^2 mse(a, b)."""
        o = self.opt
        S = float(o.size_train[0])
        k2 = (2.0 / S) ** 2
        valid = batch['valid']
        B = valid.shape[0]
        hv = valid.t()                                                                                      # [2,B]
        if F.MESH_LOSS_FUSED and not test and valid.is_cuda and batch['verts_left_gt'].shape[1] == 778 and float(S) == int(S):
            # round 5: all twelve terms below AND their weighted sum in two launches forward + one backward (csrc/loss.hip mesh_loss_*); the
            # term-by-term path that follows stays as the evaluation branch and as the restatement the fused kernels are tested against
            alpha = 0 if (epoch is None or epoch < 20) else 1
            hd = handDictList[0]
            gtp = lambda k: (batch[k % 'left'], batch[k % 'right'])
            named = dict(zip(self._ORDER, self.coefficients(0 if epoch is None else epoch)))
            mp, t = F.mesh_loss(_pair(result['verts3d']), _pair(result['verts2d']), _pair(hd['verts3d']), _pair(hd['verts2d']), _pair(paramsDict['root']),
                                {'vgt': gtp('verts_%s_gt'), 'jgt': gtp('joints_%s_gt'), 'v2gt': gtp('verts2d_%s_gt'), 'lmsgt': gtp('lms_%s_gt'),
                                 'ind': ind, 'K': batch['K_new'], 'valid': valid},
                                ((self.full_regressor_left, self.full_regressor_right), self.faces_pair, (cl.graph_perm, cr.graph_perm)),
                                int(S), getattr(o, 'down_ratio', 4), alpha != 0, [named[k] for k in F.MESH_LOSS_TERMS])
            t['_mesh_part'] = mp
            return t
        gt = lambda k: torch.stack((batch[k % 'left'], batch[k % 'right']))
        vgt, jgt, v2gt, lmsgt = gt('verts_%s_gt'), gt('joints_%s_gt'), gt('verts2d_%s_gt'), gt('lms_%s_gt')
        root_gt = jgt[:, :, 9:10]
        vgt_off = vgt - root_gt
        vp, v2p = _pair(result['verts3d']), _pair(result['verts2d'])
        regs = (self.full_regressor_left, self.full_regressor_right)

        verts2d_loss = F.rowloss(v2p, v2gt, 1, 'l2').sum() * k2                                             # :425-426
        verts_rows = F.rowloss(vp, vgt_off, 2, 'l1')                                                        # :427-428 (x valid, summed over hands below)
        jp_off = F.regress_joints_pair(*regs, vp)                                                           # :431-432
        jg_off = F.regress_joints_pair(*regs, vgt_off)
        joints_rows = F.rowloss(jp_off, jg_off, 2, 'l1')                                                    # :435-436
        alpha = 0 if (epoch is None or epoch < 20) else 1                                                   # :610 (test mode passes None)
        # edge term: weighted by alpha in `total`; while alpha == 0 it is reported but its (exactly zero) gradient is skipped
        nl, el = F.face_loss(vp, vgt_off, self.faces_pair, edge_grad=alpha != 0)                            # :452-453
        norm_loss, edge_loss = nl.sum(), el.sum()

        # GCN-level supervision: GT in GCN order, 1008 -> 252 by two pair-averagings (:461-482).
        # NB the reference feeds the LEFT GT to both hands and weights both terms by valid[:,0] (:463,481-482).
        g3 = _pool4(torch.stack((cl.vert_to_GCN(vgt_off[0]), cr.vert_to_GCN(vgt_off[0]))))
        g2 = _pool4(torch.stack((cl.vert_to_GCN(v2gt[0]), cr.vert_to_GCN(v2gt[1]))))
        hd = handDictList[0]
        gcn_rows = F.rowloss(_pair(hd['verts3d']), g3, 2, 'l1')
        gcn_2d_loss = F.rowloss(_pair(hd['verts2d']), g2, 1, 'l2').sum() * k2

        r = _pair(paramsDict['root'])                                                                       # :489-506
        root_pred = uv_root_3d(ind.t(), r[..., 1:] / 100, 0.4 + r[..., 0] / 100, batch['K_new'], int(S), getattr(o, 'down_ratio', 4))
        jp = jp_off + (root_pred if test else root_gt)
        lms = perspective(jp, batch['K_new'])
        vpred = vp + root_pred
        if test:                                                                                            # :652-653
            return tuple(t.transpose(0, 1).contiguous() for t in (vpred, jp, vgt, jgt, lms, vp, jp_off, vgt_off, jg_off))
        joints2d_loss = (F.rowloss(lms, lmsgt, 1, 'l2').unsqueeze(1) * k2 * hv).sum(0)                      # :499-500
        # The seven per-hand, per-sample terms are weighted by `valid`, summed over the hands and scaled as ONE stacked product-sum
        # ((x * valid).sum(0) [* 1000] each: 2-3 launches forward and 3-4 backward per term on the step's dependent chain;
        # `tools/probe/loss_time.py`).  NB gcn_loss weights both hands by valid[:, 0] (reference :481-482).
        rows = torch.stack((verts_rows, joints_rows, F.rowloss(root_pred, root_gt, 2, 'l1'),                # :506-507
                            F.rowloss(jp, jgt, 2, 'l1'), F.rowloss(vpred, vgt, 2, 'l1'), gcn_rows,
                            bone_direction_loss(lms.reshape(2 * B, -1, 2), lmsgt.reshape(2 * B, -1, 2), self.bone_a, self.bone_c).view(2, B)))  # :517-525
        wts = hv.unsqueeze(0).repeat(7, 1, 1)
        wts[5] = valid[:, 0]
        wts = wts * self._hand_scales
        verts_loss, joints_loss, root_loss, abs_joints_loss, abs_verts_loss, gcn_loss, bone = (rows * wts).sum(1).unbind(0)
        return {'root_loss': root_loss, 'verts_loss': verts_loss, 'abs_verts_loss': abs_verts_loss, 'verts2d_loss': verts2d_loss,
                'norm_loss': norm_loss, 'edge_loss': edge_loss, 'gcn_loss': gcn_loss, 'gcn_2d_loss': gcn_2d_loss,
                'abs_joints_loss': abs_joints_loss, 'joints2d_loss': joints2d_loss, 'joints_loss': joints_loss, 'bone_direc_loss': bone}
