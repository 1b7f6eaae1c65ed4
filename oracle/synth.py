"""TEST INFRASTRUCTURE -- deterministic weights and synthetic batches (numpy PCG64), reproducible
bit-for-bit on the build container and on the GPU box (SURVEY.md 8(c) "Weights", 8(d) inputs).

`det_state_dict(template)` fills a state_dict-shaped mapping in place from generators keyed by
crc32(parameter name): a 401 MB checkpoint cannot be a fixture, its generator can.
"""
import zlib

import numpy as np
import torch


def _rng(name, salt=0):
    return np.random.Generator(np.random.PCG64(zlib.crc32(name.encode()) + 1000003 * salt))


def det_tensor(name, shape, kind, salt=0):
    g = _rng(name, salt)
    shape = tuple(shape)
    n = int(np.prod(shape)) if len(shape) else 1
    if kind == 'weight':                       # conv / linear / embedding: N(0, 1/fan_in) keeps O(1)
        fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else shape[0]
        if name.endswith('position_embeddings.weight'):
            std = 0.5
        elif '.p3.' in name or '.p4.' in name or '.p5.' in name:   # ConvTranspose2d [Cin,Cout,k,k]
            std = (1.0 / shape[0]) ** 0.5
        else:
            std = (1.0 / fan_in) ** 0.5
        a = g.standard_normal(n, dtype=np.float32) * np.float32(std)
    elif kind == 'bias':
        a = g.standard_normal(n, dtype=np.float32) * np.float32(0.05)
    elif kind == 'gamma':                      # BN / LN / L2Norm scale
        lo, hi = (0.1, 0.3) if name.endswith('bn3.weight') else (0.6, 1.2)
        a = g.uniform(lo, hi, n).astype(np.float32)
    elif kind == 'running_mean':
        a = g.standard_normal(n, dtype=np.float32) * np.float32(0.1)
    elif kind == 'running_var':
        a = g.uniform(0.5, 1.5, n).astype(np.float32)
    else:
        raise ValueError(kind)
    return a.reshape(shape)


def classify(name, tensor):
    leaf = name.rsplit('.', 1)[-1]
    if leaf == 'num_batches_tracked':
        return 'count'
    if leaf in ('running_mean', 'running_var'):
        return leaf
    if name in ('decoder.dense_coor', 'decoder.unsample_layer.weight'):
        return 'keep'                          # real constants from gcn_core (intaghand_decoder.py:113-115,158-160)
    if leaf == 'bias':
        return 'bias'
    if tensor.dim() == 1:                      # BN / LayerNorm / L2Norm weight
        return 'gamma'
    return 'weight'


def det_state_dict(template, salt=0):
    """template: mapping name -> tensor (e.g. model.state_dict()); returns a new dict."""
    out = {}
    for k, v in template.items():
        kind = classify(k, v)
        if kind == 'count':
            out[k] = torch.zeros_like(v)
        elif kind == 'keep':
            out[k] = v.detach().clone()
        else:
            a = det_tensor(k, v.shape, kind, salt)
            if k.endswith('_l2.weight'):       # L2Norm gamma init is 10 (intaghand_encoder.py:613-616)
                a = a * np.float32(10.0)
            out[k] = torch.from_numpy(a)
    return out


def synthetic_batch(B, R=256, seed=1, variant='plain'):
    """Model inputs of SURVEY.md 8(d) / Appendix E (numpy arrays).

    variant: 'plain' | 'mixed' (10% far outliers so the ball masks fire, ~30% clouds wrap-padded
    from 300 unique points, one all-zero cloud with valid=0 when B>=2).
    """
    g = np.random.Generator(np.random.PCG64(seed))
    img = g.standard_normal((B, 3, R, R), dtype=np.float32)
    depth = g.uniform(0, 1, (B, 1, R, R)).astype(np.float32)
    xy = g.uniform(-0.1, 0.1, (B, 2, 1024, 2)).astype(np.float32)
    z = g.uniform(0.4, 0.5, (B, 2, 1024, 1)).astype(np.float32)
    cloud = np.concatenate([xy, z], -1)
    choose = g.integers(0, R * R, (B, 2, 1024), dtype=np.int64)
    ind = g.integers(0, (R // 4) ** 2, (B, 2), dtype=np.int64)
    valid = np.ones((B, 2), np.float32)
    if variant == 'mixed':
        far = g.uniform(0, 1, (B, 2, 1024)) < 0.1
        cloud[..., :2] = np.where(far[..., None], g.uniform(-0.5, 0.5, (B, 2, 1024, 2)).astype(np.float32), cloud[..., :2])
        for b in range(B):
            for h in range(2):
                if g.uniform() < 0.3:
                    reps = np.resize(np.arange(300), 1024)
                    cloud[b, h] = cloud[b, h, reps]
                    choose[b, h] = choose[b, h, reps]
        if B >= 2:
            cloud[B - 1, 1] = 0
            choose[B - 1, 1] = 0
            valid[B - 1, 1] = 0
    K = np.tile(np.array([[R, 0, R / 2], [0, R, R / 2], [0, 0, 1]], np.float32), (B, 1, 1))
    return dict(input=img, depth=depth, cloud=cloud, choose=choose, ind=ind, K_new=K, valid=valid)


DUALGRAPH_DIMS = ((63, 512, 256), (126, 256, 128), (252, 128, 64))       # (V, C_in, C_out) of DualGraph layers 0 / 1 / 2 (intaghand_decoder.py:125-139)


def dualgraph_case(level, B=3):
    """Inputs of the op_dualgraph_layer_L* fixtures: x [2 (left, right), B, V, C_in] and the output gradient gy [2, B, V, C_out]."""
    V, cin, cout = DUALGRAPH_DIMS[level]
    g = np.random.Generator(np.random.PCG64(9100 + level))
    return g.standard_normal((2, B, V, cin), dtype=np.float32), g.standard_normal((2, B, V, cout), dtype=np.float32)


def to_torch(batch, device='cpu'):
    return {k: torch.from_numpy(v).to(device) for k, v in batch.items()}


def synthetic_mano_consts(side='left', seed=7):
    """MANO-shaped constants with the real sparsity pattern sizes (the MPI-licensed pickles are not
    committed; SURVEY.md 'Licensing/size of constants')."""
    g = np.random.Generator(np.random.PCG64(seed + (0 if side == 'left' else 1)))
    c = {}
    c['v_template'] = (g.standard_normal((778, 3)) * 0.05).astype(np.float32)
    c['shapedirs'] = (g.standard_normal((778, 3, 10)) * 0.01).astype(np.float32)
    c['posedirs'] = (g.standard_normal((778, 3, 135)) * 0.002).astype(np.float32)
    J = np.zeros((16, 778), np.float32)
    for j in range(16):
        cols = g.choice(778, 118, replace=False)
        w = g.uniform(0, 1, 118).astype(np.float32)
        J[j, cols] = w / w.sum()
    c['J_regressor'] = J
    W = np.zeros((778, 16), np.float32)
    for v in range(778):
        cols = g.choice(16, g.integers(1, 7), replace=False)
        w = g.uniform(0, 1, len(cols)).astype(np.float32)
        W[v, cols] = w / w.sum()
    c['weights'] = W
    return {k: torch.from_numpy(v) for k, v in c.items()}
