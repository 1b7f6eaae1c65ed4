"""Counterparts of the reference's checkpoint helpers (lib/utils/utils.py:37-119) and of the dataset's MANO ground-truth
generation (lib/datasets/interhand.py:555-587) -- SURVEY.md 8(f) row 4."""
import torch

from . import functional as F


def load_model(model, model_path, optimizer=None, resume=False, lr=None, lr_step=None, verbose=True):
    """lib/utils/utils.py:37-98: accepts {'state_dict':...} or a bare state_dict, strips the DDP `module.` prefix, keeps
    the model's own tensor where shapes disagree, loads with strict=False; optimizer state only when resume=True."""
    ckpt = torch.load(model_path, map_location='cpu')
    sd_in = ckpt['state_dict'] if isinstance(ckpt, dict) and 'state_dict' in ckpt else ckpt
    sd = {}
    for k, v in sd_in.items():
        sd[k[7:] if k.startswith('module') and not k.startswith('module_list') else k] = v
    own = model.state_dict()
    for k in list(sd):
        if k in own:
            if tuple(sd[k].shape) != tuple(own[k].shape):
                if verbose:
                    print('Skip loading parameter {}, required shape{}, loaded shape{}.'.format(k, tuple(own[k].shape), tuple(sd[k].shape)))
                sd[k] = own[k]
        elif verbose:
            print('Drop parameter {}.'.format(k))
    model.load_state_dict(sd, strict=False)
    start_epoch = 0
    if optimizer is not None and resume:
        if 'optimizer' in ckpt:
            optimizer.load_state_dict(ckpt['optimizer'])       # torch.optim.Adam layout on both sides (FlatAdam.state_dict)
            start_epoch = ckpt['epoch']
            if lr is not None:                                  # lib/utils/utils.py:88-94; lr=None keeps the checkpoint's rate
                start_lr = lr
                for step in (lr_step or []):
                    if start_epoch >= step:
                        start_lr *= 0.1
                for group in optimizer.param_groups:
                    group['lr'] = start_lr
        elif verbose:
            print('No optimizer parameters in checkpoint.')
    return (model, optimizer, start_epoch) if optimizer is not None else model


def save_model(path, epoch, model, optimizer=None):
    """lib/utils/utils.py:101-119: {'epoch', 'state_dict'[, 'optimizer']} with plain contiguous (OIHW) tensors so the
    file loads into the reference model as well."""
    m = model.module if hasattr(model, 'module') else model
    data = {'epoch': epoch, 'state_dict': {k: v.detach().cpu().contiguous() for k, v in m.state_dict().items()}}
    if optimizer is not None:
        data['optimizer'] = optimizer.state_dict()
    torch.save(data, path)


def mano_gt_from_coeff(mano_consts, mano_coeff, K):
    """H2O ground truth from `mano_coeff` [B,124] (per hand 62 floats: valid, trans 1:4, rot 4:7, pose 7:52, shape 52:62;
    left then right) as lib/datasets/interhand.py:555-587 does per sample on the CPU -- here batched on the GPU with the
    LBS kernel.  mano_consts: {'left': consts, 'right': consts}; K [B,3,3].  Returns per hand verts3d/joints3d/verts2d/joints2d."""
    out = {}
    for hi, hand in enumerate(('left', 'right')):
        p = mano_coeff[:, 62 * hi:62 * (hi + 1)]
        v, j = F.mano_lbs(mano_consts[hand], p[:, 4:7].contiguous(), p[:, 7:52].contiguous(), p[:, 52:62].contiguous(),
                          trans=p[:, 1:4].contiguous(), side=hand)
        proj = lambda x: (x @ K.transpose(1, 2))[..., :2] / (x @ K.transpose(1, 2))[..., 2:]
        out[hand] = {'verts3d': v, 'joints3d': j, 'verts2d': proj(v), 'joints2d': proj(j)}
    return out


def fix_shape(mano_consts):
    """lib/datasets/interhand.py:120-123: the released MANO_LEFT.pkl carries the right hand's x-sign in its shape blend
    shapes; if left and right `shapedirs[:, 0, :]` (x rows, [778,3,10]) are (nearly) identical, flip the left one in place.
    The reference applies this to the loss module's layers (simplified.py:52) but not to the dataset's (interhand.py:460-461)
    nor to ManoRender's (Mano_render.py:61-66) -- callers choose, as there.  Returns True when it flipped."""
    l, r = mano_consts['left']['shapedirs'], mano_consts['right']['shapedirs']
    if float(torch.sum(torch.abs(l[:, 0, :] - r[:, 0, :]))) < 1:
        l[:, 0, :] *= -1
        return True
    return False


def mano_from_params(mano_consts, params_map, ind, K, input_res, down_ratio=4):
    """The legacy MANO branch of the loss (simplified.py:730-736): decode the 122-channel params head at the two centre pixels
    (`Split_coeff`) and run both MANO layers (no translation, like the reference call) -> verts [2,B,778,3], joints [2,B,21,3], trans."""
    orient, pose, shape, trans = F.mano_split_coeff(params_map, ind, K, input_res, down_ratio)
    out = [F.mano_lbs(mano_consts[h], orient[i], pose[i], shape[i], None, side=h) for i, h in enumerate(('left', 'right'))]
    return torch.stack([o[0] for o in out]), torch.stack([o[1] for o in out]), trans


def load_mano_constants(npz_path, device=None):
    """`mano_constants.npz` (written at the user's site by tools/convert_mano.py from MANO_LEFT.pkl / MANO_RIGHT.pkl) ->
    (loss_consts, lbs_consts): what the reference builds in `ManoLayer.__init__` (lib/models/networks/manolayer.py:100-160),
    `ManoModel.process_J_regressor` (lib/models/hand3d/Mano_model.py:309-323) and `fix_shape` (lib/datasets/interhand.py:120-123).

      loss_consts : {'full_regressor_left/right' [21,778] f32, 'faces_left/right' [1538,3] i64}        -> CtdetLoss(opt, loss_consts)
      lbs_consts  : {'left' / 'right': {'v_template', 'shapedirs', 'posedirs', 'J_regressor', 'weights'}}  -> F.mano_lbs(lbs_consts[side], ...)
    """
    import numpy as np
    z = np.load(npz_path)
    need = [k + '_' + s for s in ('left', 'right') for k in ('v_template', 'shapedirs', 'posedirs', 'J_regressor', 'weights', 'faces', 'full_regressor')]
    missing = [k for k in need if k not in z.files]
    if missing:
        raise KeyError("load_mano_constants: %s lacks %s (was it written by tools/convert_mano.py?)" % (npz_path, missing))
    shapes = {'v_template': (778, 3), 'shapedirs': (778, 3, 10), 'posedirs': (778, 3, 135), 'J_regressor': (16, 778), 'weights': (778, 16),
              'faces': (1538, 3), 'full_regressor': (21, 778)}
    for k in need:
        if tuple(z[k].shape) != shapes[k.rsplit('_', 1)[0]]:
            raise ValueError("load_mano_constants: %s has shape %s" % (k, tuple(z[k].shape)))
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device) if device is not None else torch.from_numpy(np.ascontiguousarray(a))
    loss_consts, lbs_consts = {}, {}
    for s in ('left', 'right'):
        loss_consts['full_regressor_' + s] = t(z['full_regressor_' + s].astype(np.float32))
        loss_consts['faces_' + s] = t(z['faces_' + s].astype(np.int64))
        lbs_consts[s] = {k: t(z[k + '_' + s].astype(np.float32)) for k in ('v_template', 'shapedirs', 'posedirs', 'J_regressor', 'weights')}
    return loss_consts, lbs_consts
