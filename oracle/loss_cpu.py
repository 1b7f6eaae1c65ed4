"""TEST INFRASTRUCTURE -- CPU restatement of the reference's live loss branch, `CtdetLoss.forward`
(lib/trains/simplified.py:364-655, H2O flags of scripts/train.sh: --reproj_loss --bone_loss), in plain PyTorch,
hand by hand in the reference's own order.  Works in float32 and float64.

Pinned by tests/golden/loss_ctdet_B3_R256.npz (outputs of the reference's own CtdetLoss run through
oracle/ref_harness.py): tests/test_oracle_vs_golden.py::test_loss_oracle_matches_reference_golden.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import torch
import torch.nn.functional as TF

# bone end points of get_bone_loss' 20x21 incidence matrix (lib/models/losses.py:34-53)
BONES = [(0, 1), (1, 2), (2, 3), (3, 4), (0, 5), (5, 6), (6, 7), (7, 8), (0, 9), (9, 10), (10, 11), (11, 12),
         (0, 13), (13, 14), (14, 15), (15, 16), (0, 17), (17, 18), (18, 19), (19, 20)]


def sigmoid_clamped(x):
    """lib/models/utils.py:8-10 `_sigmoid`."""
    return torch.clamp(torch.sigmoid(x), min=1e-4, max=1 - 1e-4)


def neg_loss(pred, gt):
    """lib/models/losses.py:138-165 `_neg_loss` -> [B]."""
    pos = gt.eq(1).to(pred.dtype)
    neg = gt.lt(1).to(pred.dtype)
    pos_loss = (torch.log(pred) * torch.pow(1 - pred, 2) * pos).sum(dim=(1, 2, 3))
    neg_l = (torch.log(1 - pred) * torch.pow(pred, 2) * torch.pow(1 - gt, 4) * neg).sum(dim=(1, 2, 3))
    num_pos = pos.sum(dim=(1, 2, 3))
    if num_pos.sum() == 0:
        return -neg_l
    return -(pos_loss + neg_l) / (num_pos + 1e-3)


def nms_topk1(heat):
    """intaghand_encoder.py:349-367: 5x5 max-pool NMS then top-1 of ONE channel -> index [B,1]."""
    hmax = TF.max_pool2d(heat, (5, 5), stride=1, padding=2)
    keep = (hmax == heat).to(heat.dtype)
    b = heat.shape[0]
    return torch.topk((heat * keep).view(b, -1), 1)[1]


def normal_loss(pred, gt, face):
    """simplified.py:66-92."""
    n = lambda v: TF.normalize(v, p=2, dim=2)
    v1o = n(pred[:, face[:, 1]] - pred[:, face[:, 0]])
    v2o = n(pred[:, face[:, 2]] - pred[:, face[:, 0]])
    v3o = n(pred[:, face[:, 2]] - pred[:, face[:, 1]])
    v1g = n(gt[:, face[:, 1]] - gt[:, face[:, 0]])
    v2g = n(gt[:, face[:, 2]] - gt[:, face[:, 0]])
    ng = n(torch.cross(v1g, v2g, dim=2))
    cos = [torch.abs(torch.sum(v * ng, 2, keepdim=True)) for v in (v1o, v2o, v3o)]
    return torch.cat(cos, 1).mean()


def edge_length_loss(pred, gt, face):
    """simplified.py:95-115."""
    d = lambda x, a, b: torch.sqrt(torch.sum((x[:, face[:, a]] - x[:, face[:, b]]) ** 2, 2, keepdim=True))
    diffs = [torch.abs(d(pred, a, b) - d(gt, a, b)) for a, b in ((0, 1), (0, 2), (1, 2))]
    return torch.cat(diffs, 1).mean()


def mesh_downsample(feat):
    """simplified.py:117-122 (AvgPool1d(2) along the vertex axis)."""
    return TF.avg_pool1d(feat.permute(0, 2, 1), 2).permute(0, 2, 1)


def bone_loss(j2d, gt2d):
    """lib/models/losses.py:26-94 `get_bone_loss` with unit confidences -> [B]."""
    a = torch.tensor([b[0] for b in BONES])
    c = torch.tensor([b[1] for b in BONES])

    def unit(j):
        v = j[:, c] - j[:, a]
        return v / torch.sqrt((v ** 2).sum(-1, keepdim=True) + 1e-4)
    return ((unit(j2d) - unit(gt2d)) ** 2).sum(-1).mean(dim=1)


def landmarks(pts, K):
    """Mano_render.py:203-209 `get_Landmarks_new`."""
    p = pts.bmm(K.reshape(-1, 3, 3).transpose(2, 1))
    return p[..., :2] / (p[..., 2:] + 1e-7)


def uv_root_3d(index, off, depth, K, input_res, down_ratio):
    """Mano_render.py:211-223 `get_uv_root_3d`; index [B,1]."""
    g = input_res // down_ratio
    cx = (index % g) * down_ratio
    cy = (index // g) * down_ratio
    x = depth * (off[:, 0] + cx.squeeze(1) - K[:, 0, 2]) / (K[:, 0, 0] + 1e-7)
    y = depth * (off[:, 1] + cy.squeeze(1) - K[:, 1, 2]) / (K[:, 1, 1] + 1e-7)
    return torch.stack((x, y, depth), 1).unsqueeze(1)


class Converter:
    """GCN_vert_convert (intaghand_decoder.py:32-43): 778 MANO vertices <-> 1008 graph nodes."""

    def __init__(self, perm, perm_reverse):
        self.perm = torch.as_tensor(perm).long()
        self.rev = torch.as_tensor(perm_reverse).long()[:778]

    def vert_to_GCN(self, x):
        return x[:, self.perm]

    def GCN_to_vert(self, x):
        return x[:, self.rev]


def ctdet_loss(opt, consts, result, paramsDict, handDictList, otherInfo, batch, mode, epoch):
    """-> (loss [B], stats) in train mode, the evaluation 9-tuple (simplified.py:652-653) in 'val' / 'test'.
    consts: {'full_regressor_left/right' [21,778], 'faces_left/right' [1538,3]}; otherInfo carries converter_left/right
    (objects with vert_to_GCN)."""
    S = opt.size_train[0]
    B = batch['joints_left_gt'].size(0)
    dt = result['verts3d']['left'].dtype
    valid = batch['valid'].to(dt)
    test = mode in ('val', 'test')
    nrm = lambda x: x / S * 2 - 1
    mask_loss = TF.smooth_l1_loss(otherInfo['mask'], batch['mask'].to(dt))                                  # :368
    hms_loss = TF.mse_loss(otherInfo['hms'], batch['hms'].to(dt))                                           # :374
    center_hm = sigmoid_clamped(otherInfo['ret']['hm'])                                                     # :376
    if test:                                                                                                # :378-384
        ch = center_hm.detach()
        ind_l, ind_r = nms_topk1(ch[:, :1]), nms_topk1(ch[:, 1:])
    else:
        ind_l, ind_r = batch['ind'][:, :1], batch['ind'][:, 1:]
    hm_loss = neg_loss(center_hm, batch['hm'].to(dt))                                                       # :391
    G = {h: {'v': batch['verts_%s_gt' % h].to(dt), 'j': batch['joints_%s_gt' % h].to(dt), 'v2': batch['verts2d_%s_gt' % h].to(dt),
             'lms': batch['lms_%s_gt' % h].to(dt)} for h in ('left', 'right')}
    P, reg, faces = {}, {}, {}
    for hi, h in enumerate(('left', 'right')):
        g = G[h]
        g['root'] = g['j'][:, 9:10]
        g['v_off'] = g['v'] - g['root']
        reg[h] = consts['full_regressor_' + h].to(dt)
        faces[h] = consts['faces_' + h].long()
        P[h] = {'v_off': result['verts3d'][h]}
    verts2d_loss = TF.mse_loss(nrm(result['verts2d']['left']), nrm(G['left']['v2'])) + \
        TF.mse_loss(nrm(result['verts2d']['right']), nrm(G['right']['v2']))                                 # :425-426
    l1 = lambda a, b: (a - b).abs().reshape(B, -1).mean(dim=1)
    verts_loss = l1(P['left']['v_off'], G['left']['v_off']) * valid[:, 0] + l1(P['right']['v_off'], G['right']['v_off']) * valid[:, 1]
    for h in ('left', 'right'):                                                                              # :431-434
        P[h]['j_off'] = torch.matmul(reg[h], P[h]['v_off'])
        G[h]['j_off'] = torch.matmul(reg[h], G[h]['v_off'])
    joints_loss = l1(P['left']['j_off'], G['left']['j_off']) * valid[:, 0] + l1(P['right']['j_off'], G['right']['j_off']) * valid[:, 1]
    norm_loss = sum(normal_loss(P[h]['v_off'], G[h]['v_off'], faces[h]) for h in ('left', 'right'))         # :452
    edge_loss = sum(edge_length_loss(P[h]['v_off'], G[h]['v_off'], faces[h]) for h in ('left', 'right'))    # :453
    # GCN-level supervision (:461-482).  NB the reference feeds the LEFT ground truth to both hands and weights both
    # 3-D terms by valid[:, 0].
    cl, cr = otherInfo['converter_left'], otherInfo['converter_right']
    pool4 = lambda x: mesh_downsample(mesh_downsample(x))
    g3 = {'left': pool4(cl.vert_to_GCN(G['left']['v_off'])), 'right': pool4(cr.vert_to_GCN(G['left']['v_off']))}
    g2 = {'left': pool4(cl.vert_to_GCN(G['left']['v2'])), 'right': pool4(cr.vert_to_GCN(G['right']['v2']))}
    hd = handDictList[0]
    gcn_loss = l1(hd['verts3d']['left'], g3['left']) * valid[:, 0] + l1(hd['verts3d']['right'], g3['right']) * valid[:, 0]
    gcn_2d_loss = TF.mse_loss(nrm(hd['verts2d']['left']), nrm(g2['left'])) + TF.mse_loss(nrm(hd['verts2d']['right']), nrm(g2['right']))
    K = batch['K_new'].to(dt)
    down = getattr(opt, 'down_ratio', 4)
    for h, ind in (('left', ind_l), ('right', ind_r)):                                                      # :489-506
        r = paramsDict['root'][h]
        P[h]['root'] = uv_root_3d(ind, r[:, 1:] / 100, 0.4 + r[:, 0] / 100, K, int(S), down)
        P[h]['j'] = P[h]['j_off'] + (P[h]['root'] if test else G[h]['root'])
        P[h]['lms'] = landmarks(P[h]['j'], K)
        P[h]['v'] = P[h]['v_off'] + P[h]['root']
    joints2d_loss = TF.mse_loss(nrm(P['left']['lms']), nrm(G['left']['lms'])) * valid[:, 0] + \
        TF.mse_loss(nrm(P['right']['lms']), nrm(G['right']['lms'])) * valid[:, 1]
    root_loss = l1(P['left']['root'], G['left']['root']) * valid[:, 0] * 1000 + l1(P['right']['root'], G['right']['root']) * valid[:, 1] * 1000
    abs_joints_loss = (l1(P['left']['j'], G['left']['j']) * valid[:, 0] + l1(P['right']['j'], G['right']['j']) * valid[:, 1]) * 1000
    abs_verts_loss = (l1(P['left']['v'], G['left']['v']) * valid[:, 0] + l1(P['right']['v'], G['right']['v']) * valid[:, 1]) * 1000
    bone = bone_loss(P['left']['lms'], G['left']['lms']) * valid[:, 0] + bone_loss(P['right']['lms'], G['right']['lms']) * valid[:, 1]
    if test:                                                                                                # :598-607, :652-653
        both = lambda k, src: torch.stack((src['left'][k], src['right'][k]), 1)
        return (both('v', P), both('j', P), both('v', G), both('j', G), both('lms', P), both('v_off', P), both('j_off', P),
                both('v_off', G), both('j_off', G))
    alpha = 0 if epoch < 20 else 1                                                                          # :610
    w = getattr(opt, 'reproj_weight', 1.0)
    loss = getattr(opt, 'center_weight', 200.0) * hm_loss + w * root_loss
    loss = loss + w * verts_loss * 500 + w * abs_verts_loss * 0.1 + w * verts2d_loss * 50 + w * norm_loss * 10
    loss = loss + w * edge_loss * 2000 * alpha + w * gcn_loss * 100 + w * gcn_2d_loss * 50
    loss = loss + w * mask_loss * 2000 + w * abs_joints_loss * 0.1 + w * hms_loss * 2000
    loss = loss + w * joints2d_loss * 1000 * alpha + w * joints_loss * 500
    loss = loss + getattr(opt, 'bone_dir_weight', 200.0) * bone
    stats = {'hm_loss': hm_loss, 'root_loss': root_loss, 'verts_loss': verts_loss, 'abs_verts_loss': abs_verts_loss,
             'verts2d_loss': verts2d_loss, 'norm_loss': norm_loss, 'edge_loss': edge_loss, 'gcn_loss': gcn_loss,
             'gcn_2d_loss': gcn_2d_loss, 'mask_loss': mask_loss, 'abs_joints_loss': abs_joints_loss, 'hms_loss': hms_loss,
             'joints2d_loss': joints2d_loss, 'joints_loss': joints_loss, 'bone_direc_loss': bone, 'loss': loss}
    return loss, stats


def evaluation_metrics(tup, lms_gt):
    """The H2O branch of `BaseTrainer.evaluation` (lib/trains/base_trainer.py:244-323) for ONE batch: mean Euclidean
    errors in mm per hand, absolute and root-relative, and the 2-D landmark error in pixels.  tup = the loss module's
    test-mode 9-tuple; lms_gt = (lms_left_gt, lms_right_gt).  The reference sums these per batch and divides by the
    number of batches (:420-429)."""
    vp, jp, vg, jg, lms, vpo, jpo, vgo, jgo = tup
    e = lambda a, b, hand: float(torch.norm(a[:, hand] - b[:, hand], dim=-1).mean()) * 1000
    out = {}
    for hi, h in enumerate(('left', 'right')):
        out['abs_%s_joints' % h] = e(jp, jg, hi)
        out['abs_%s_verts' % h] = e(vp, vg, hi)
        out['off_%s_joints' % h] = e(jpo, jgo, hi)
        out['off_%s_verts' % h] = e(vpo, vgo, hi)
    out['lms_px'] = (float(torch.norm(lms[:, 0] - lms_gt[0], dim=-1).mean()) + float(torch.norm(lms[:, 1] - lms_gt[1], dim=-1).mean())) / 2
    return out
