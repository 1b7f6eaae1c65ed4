"""Counterpart of the reference's live loss branch, `CtdetLoss.forward` (lib/trains/simplified.py:364-655),
for the H2O RGB-D two-hand task (flags of scripts/train.sh: --reproj_loss --bone_loss, dataset H2O).

SURVEY.md 8(f) row 1 ("next"): the loss is ~40 launch-bound element-wise / tiny-matmul ops.  In this round
it is sync-free device code written with aten element-wise ops plus the HIP joint-regressor kernel, so the
whole train step is hipGraph-capturable; fusing it into a handful of HIP kernels is the next widening step.
Differences from the reference are host-side only: no `.cpu()` debug dumps (simplified.py:527-596) and the
focal loss' `if num_pos.sum() == 0` host branch (lib/models/losses.py:161) is a device-side select.
"""
import torch
import torch.nn as nn
import torch.nn.functional as TF

from .. import functional as F

_BONES = [(0, 1), (1, 2), (2, 3), (3, 4), (0, 5), (5, 6), (6, 7), (7, 8), (0, 9), (9, 10), (10, 11), (11, 12),
          (0, 13), (13, 14), (14, 15), (15, 16), (0, 17), (17, 18), (18, 19), (19, 20)]      # losses.py:34-53


def sigmoid_clamped(x):
    """lib/models/utils.py:8-10 `_sigmoid` (out-of-place: the caller's logits stay intact)."""
    return torch.clamp(torch.sigmoid(x), min=1e-4, max=1 - 1e-4)


def focal_loss(pred, gt):
    """lib/models/losses.py:138-165 `_neg_loss`, per-sample [B]."""
    pos = gt.eq(1).float()
    neg = gt.lt(1).float()
    negw = torch.pow(1 - gt, 4)
    pos_loss = (torch.log(pred) * torch.pow(1 - pred, 2) * pos).sum(dim=(1, 2, 3))
    neg_loss = (torch.log(1 - pred) * torch.pow(pred, 2) * negw * neg).sum(dim=(1, 2, 3))
    num_pos = pos.sum(dim=(1, 2, 3))
    return torch.where(num_pos.sum() == 0, -neg_loss, -(pos_loss + neg_loss) / (num_pos + 1e-3))


def bone_direction_loss(j2d, gt2d, a, c):
    """lib/models/losses.py:26-94 `get_bone_loss` with unit confidences -> [B]; a, c = bone end-point indices."""

    def unit(j):
        v = j[:, c] - j[:, a]                                            # [B,20,2]
        return v / torch.sqrt((v ** 2).sum(-1, keepdim=True) + 1e-4)
    return ((unit(j2d) - unit(gt2d)) ** 2).sum(-1).mean(dim=1)


def _unit(v):
    return TF.normalize(v, p=2, dim=2)


def normal_loss(pred, gt, face):
    """simplified.py:66-91."""
    f0, f1, f2 = face[:, 0], face[:, 1], face[:, 2]
    v1o, v2o, v3o = _unit(pred[:, f1] - pred[:, f0]), _unit(pred[:, f2] - pred[:, f0]), _unit(pred[:, f2] - pred[:, f1])
    n = _unit(torch.cross(_unit(gt[:, f1] - gt[:, f0]), _unit(gt[:, f2] - gt[:, f0]), dim=2))
    cos = [torch.abs((v * n).sum(2, keepdim=True)) for v in (v1o, v2o, v3o)]
    return torch.cat(cos, 1).mean()


def edge_length_loss(pred, gt, face):
    """simplified.py:94-115."""
    f0, f1, f2 = face[:, 0], face[:, 1], face[:, 2]

    def d(x, i, j):
        return torch.sqrt(((x[:, i] - x[:, j]) ** 2).sum(2, keepdim=True))
    diffs = [torch.abs(d(pred, i, j) - d(gt, i, j)) for i, j in ((f0, f1), (f0, f2), (f1, f2))]
    return torch.cat(diffs, 1).mean()


def perspective(pts, K):
    """Mano_render.py:203-209 `get_Landmarks_new`."""
    p = pts.bmm(K.reshape(-1, 3, 3).transpose(2, 1))
    return p[..., :2] / (p[..., 2:] + 1e-7)


def uv_root_3d(index, root_xy, root_z, K, input_res, down_ratio=4):
    """Mano_render.py:211-223 `get_uv_root_3d`."""
    g = input_res // down_ratio
    cx = ((index % g) * down_ratio).squeeze(-1)
    cy = ((index // g) * down_ratio).squeeze(-1)
    x = root_z * (root_xy[:, 0] + cx - K[:, 0, 2]) / (K[:, 0, 0] + 1e-7)
    y = root_z * (root_xy[:, 1] + cy - K[:, 1, 2]) / (K[:, 1, 1] + 1e-7)
    return torch.stack((x, y, root_z), 1).unsqueeze(1)


def _pool2(x):
    """simplified.py:117-122 `mesh_downsample` (average pairs along the vertex axis)."""
    B, N, Fd = x.shape
    return x.view(B, N // 2, 2, Fd).mean(2)


def projection_batch(scale, trans2d, pts, img_size):
    s = (scale * img_size).view(-1, 1, 1)
    t = (trans2d * img_size / 2 + img_size / 2).unsqueeze(1)
    return s * pts[..., :2] + t


class CtdetLoss(nn.Module):
    """opt fields used: size_train, center_weight (200), reproj_weight (1), bone_dir_weight (200), down_ratio (4).
    `consts`: {'full_regressor_left/right' [21,778], 'faces_left/right' [1538,3] int64}."""

    def __init__(self, opt, consts):
        super().__init__()
        self.opt = opt
        for k, v in consts.items():
            self.register_buffer(k, v.clone(), persistent=False)
        self.register_buffer('bone_a', torch.tensor([b[0] for b in _BONES]), persistent=False)
        self.register_buffer('bone_c', torch.tensor([b[1] for b in _BONES]), persistent=False)

    def forward(self, result, paramsDict, handDictList, otherInfo, batch, mode, epoch):
        o = self.opt
        S = float(o.size_train[0])
        valid = batch['valid']
        B = valid.shape[0]
        l1 = lambda a, b: TF.l1_loss(a, b, reduction='none').reshape(B, -1).mean(dim=1)
        l2 = TF.mse_loss
        nrm = lambda x: x / S * 2 - 1

        test = mode in ('val', 'test')
        if test:
            from ..networks.intaghand_encoder import nms_top1_centers
            ind = nms_top1_centers(sigmoid_clamped(otherInfo['ret']['hm']))                # :376-389
        else:
            ind = batch['ind']
        ind_l, ind_r = ind[:, :1], ind[:, 1:]

        vgt = {'left': batch['verts_left_gt'], 'right': batch['verts_right_gt']}
        jgt = {'left': batch['joints_left_gt'], 'right': batch['joints_right_gt']}
        v2gt = {'left': batch['verts2d_left_gt'], 'right': batch['verts2d_right_gt']}
        lmsgt = {'left': batch['lms_left_gt'], 'right': batch['lms_right_gt']}
        root_gt = {h: jgt[h][:, 9:10] for h in jgt}
        vgt_off = {h: vgt[h] - root_gt[h] for h in vgt}
        vpred_off = result['verts3d']
        reg = {'left': self.full_regressor_left, 'right': self.full_regressor_right}
        face = {'left': self.faces_left, 'right': self.faces_right}
        hv = {'left': valid[:, 0], 'right': valid[:, 1]}

        verts2d_loss = sum(l2(nrm(result['verts2d'][h]), nrm(v2gt[h])) for h in ('left', 'right'))          # :425-426
        verts_loss = sum(l1(vpred_off[h], vgt_off[h]) * hv[h] for h in ('left', 'right'))                   # :427-428
        jpred_off = {h: F.regress_joints(reg[h], vpred_off[h]) for h in ('left', 'right')}                  # :431-432
        jgt_off = {h: F.regress_joints(reg[h], vgt_off[h]) for h in ('left', 'right')}
        joints_loss = sum(l1(jpred_off[h], jgt_off[h]) * hv[h] for h in ('left', 'right'))                  # :435-436
        norm_loss = sum(normal_loss(vpred_off[h], vgt_off[h], face[h]) for h in ('left', 'right'))          # :452
        alpha = 0 if epoch < 20 else 1                                                                      # :610
        with torch.set_grad_enabled(torch.is_grad_enabled() and alpha != 0):
            # weighted by alpha below: while alpha == 0 the term is reported but contributes an exactly-zero gradient,
            # so its backward graph is not built
            edge_loss = sum(edge_length_loss(vpred_off[h], vgt_off[h], face[h]) for h in ('left', 'right'))  # :453

        # GCN-level supervision: GT in GCN order, 1008 -> 252 by two pair-averagings (:461-482).
        # NB the reference feeds the LEFT GT to both hands and weights both terms by valid[:,0] (:463,481-482).
        cl, cr = otherInfo['converter_left'], otherInfo['converter_right']
        g3 = {'left': _pool2(_pool2(cl.vert_to_GCN(vgt_off['left']))), 'right': _pool2(_pool2(cr.vert_to_GCN(vgt_off['left'])))}
        g2 = {'left': _pool2(_pool2(cl.vert_to_GCN(v2gt['left']))), 'right': _pool2(_pool2(cr.vert_to_GCN(v2gt['right'])))}
        hd = handDictList[0]
        gcn_loss = sum(l1(hd['verts3d'][h], g3[h]) * valid[:, 0] for h in ('left', 'right'))
        gcn_2d_loss = sum(l2(nrm(hd['verts2d'][h]), nrm(g2[h])) for h in ('left', 'right'))

        root_pred, jpred, vpred, lms_proj = {}, {}, {}, {}
        for h, idx in (('left', ind_l), ('right', ind_r)):                                                  # :489-506
            r = paramsDict['root'][h]
            root_pred[h] = uv_root_3d(idx, r[:, 1:] / 100, 0.4 + r[:, 0] / 100, batch['K_new'], int(S), getattr(o, 'down_ratio', 4))
            jpred[h] = jpred_off[h] + (root_pred[h] if test else root_gt[h])
            lms_proj[h] = perspective(jpred[h], batch['K_new'])
            vpred[h] = vpred_off[h] + root_pred[h]
        if test:                                                                                            # :652-653
            cat = lambda d, n: torch.cat((d['left'], d['right']), dim=1).reshape(B, -1, n, d['left'].shape[-1])
            return (cat(vpred, 778), cat(jpred, 21), cat(vgt, 778), cat(jgt, 21), cat(lms_proj, 21),
                    cat(vpred_off, 778), cat(jpred_off, 21), cat(vgt_off, 778), cat(jgt_off, 21))
        joints2d_loss = sum(l2(nrm(lms_proj[h]), nrm(lmsgt[h])) * hv[h] for h in ('left', 'right'))         # :499-500
        root_loss = sum(l1(root_pred[h], root_gt[h]) * hv[h] * 1000 for h in ('left', 'right'))             # :506-507
        abs_joints_loss = sum(l1(jpred[h], jgt[h]) * hv[h] for h in ('left', 'right')) * 1000
        abs_verts_loss = sum(l1(vpred[h], vgt[h]) * hv[h] for h in ('left', 'right')) * 1000
        bone = sum(bone_direction_loss(lms_proj[h], lmsgt[h], self.bone_a, self.bone_c) * hv[h] for h in ('left', 'right'))           # :517-525

        # dense-map terms (created last: their backward is issued first, ahead of the launch-bound mesh terms)
        mask_loss = TF.smooth_l1_loss(otherInfo['mask'], batch['mask'])                    # :368
        hms_loss = l2(otherInfo['hms'], batch['hms'])                                      # :374
        hm_loss = focal_loss(sigmoid_clamped(otherInfo['ret']['hm']), batch['hm'])         # :376, :391
        w = getattr(o, 'reproj_weight', 1.0)
        loss = getattr(o, 'center_weight', 200.0) * hm_loss + w * root_loss
        loss = loss + w * (verts_loss * 500 + abs_verts_loss * 0.1 + verts2d_loss * 50 + norm_loss * 10 +
                           edge_loss * 2000 * alpha + gcn_loss * 100 + gcn_2d_loss * 50)
        loss = loss + w * (mask_loss * 2000 + abs_joints_loss * 0.1 + hms_loss * 2000 +
                           joints2d_loss * 1000 * alpha + joints_loss * 500)
        loss = loss + getattr(o, 'bone_dir_weight', 200.0) * bone
        stats = {'hm_loss': hm_loss, 'root_loss': root_loss, 'verts_loss': verts_loss, 'abs_verts_loss': abs_verts_loss,
                 'verts2d_loss': verts2d_loss, 'norm_loss': norm_loss, 'edge_loss': edge_loss, 'gcn_loss': gcn_loss,
                 'gcn_2d_loss': gcn_2d_loss, 'mask_loss': mask_loss, 'abs_joints_loss': abs_joints_loss, 'hms_loss': hms_loss,
                 'joints2d_loss': joints2d_loss, 'joints_loss': joints_loss, 'bone_direc_loss': bone, 'loss': loss}
        return loss, stats, None, None
