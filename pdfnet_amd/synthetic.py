"""Synthetic H2O-shaped training batches and MANO-shaped constants (the dataset counterpart for benchmarks
and smoke tests: the H2O data and the MPI-licensed MANO pickles are not available/redistributable).

Batch schema = what `InterHandDataset.__getitem__` emits after collation (reference
lib/datasets/interhand.py:983-1023); value ranges per SURVEY.md 8(d).
"""
import os

import numpy as np
import torch

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "gcn_core.npz")
MANO_NEW_ORDER = [0, 13, 14, 15, 16, 1, 2, 3, 17, 4, 5, 6, 18, 10, 11, 12, 19, 7, 8, 9, 20]


def synthetic_loss_constants(seed=7):
    """full_regressor-shaped [21,778] matrices (16 sparse joint rows of 118 non-zeros + 5 one-hot tip rows
    745/317/444/556/673, reordered as Mano_model.py:309-323) and the mesh faces shipped with the graphs."""
    z = np.load(_DATA)
    out = {}
    for hi, hand in enumerate(('left', 'right')):
        g = np.random.Generator(np.random.PCG64(seed + hi))
        J = np.zeros((21, 778), np.float32)
        for j in range(16):
            cols = g.choice(778, 118, replace=False)
            w = g.uniform(0, 1, 118).astype(np.float32)
            J[j, cols] = w / w.sum()
        for r, v in enumerate((745, 317, 444, 556, 673)):
            J[16 + r, v] = 1.0
        out['full_regressor_' + hand] = torch.from_numpy(J[MANO_NEW_ORDER].copy())
        out['faces_' + hand] = torch.from_numpy(z['mesh_faces_' + hand].astype(np.int64))
    return out


def synthetic_train_batch(B, R=256, seed=1, consts=None):
    """dict of CPU tensors with every key the model + loss read."""
    consts = consts or synthetic_loss_constants()
    g = np.random.Generator(np.random.PCG64(seed))
    f32 = np.float32
    b = {}
    b['input'] = g.standard_normal((B, 3, R, R), dtype=f32)
    b['depth'] = g.uniform(0, 1, (B, 1, R, R)).astype(f32)
    b['cloud'] = np.concatenate([g.uniform(-0.1, 0.1, (B, 2, 1024, 2)), g.uniform(0.4, 0.5, (B, 2, 1024, 1))], -1).astype(f32)
    b['choose'] = g.integers(0, R * R, (B, 2, 1024), dtype=np.int64)
    b['ind'] = g.integers(0, (R // 4) ** 2, (B, 2), dtype=np.int64)
    b['K_new'] = np.tile(np.array([[R, 0, R / 2], [0, R, R / 2], [0, 0, 1]], f32), (B, 1, 1))
    b['valid'] = np.ones((B, 2), f32)
    hm = g.uniform(0, 0.9, (B, 2, R // 4, R // 4)).astype(f32)
    for i in range(B):
        for c in range(2):
            hm[i, c].flat[b['ind'][i, c]] = 1.0                       # one exact positive per channel
    b['hm'] = hm
    b['hms'] = g.uniform(0, 1, (B, 42, R // 4, R // 4)).astype(f32)
    b['mask'] = (g.uniform(0, 1, (B, 2, R, R)) < 0.5).astype(f32)
    out = {k: torch.from_numpy(v) for k, v in b.items()}
    K = out['K_new']
    for hi, hand in enumerate(('left', 'right')):
        v = torch.from_numpy(np.concatenate([g.uniform(-0.1, 0.1, (B, 778, 2)), g.uniform(0.4, 0.5, (B, 778, 1))], -1).astype(f32))
        j = torch.einsum('jv,bvc->bjc', consts['full_regressor_' + hand], v)
        proj = lambda p: (p @ K.transpose(1, 2))[..., :2] / (p @ K.transpose(1, 2))[..., 2:]
        out['verts_%s_gt' % hand] = v
        out['joints_%s_gt' % hand] = j
        out['verts2d_%s_gt' % hand] = proj(v)
        out['lms_%s_gt' % hand] = proj(j)
    return out


def to_device(batch, device):
    return {k: v.to(device, non_blocking=True) for k, v in batch.items()}
