"""Shared helpers for the parity tests (test infrastructure; may import oracle/)."""
import os
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def gold(name):
    z = np.load(os.path.join(GOLD, name + ".npz"))
    return {k: z[k] for k in z.files}


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def make_opt(R=256, **over):
    """The `opt` fields the model reads (reference lib/opts.py:221-239, heads :291-295)."""
    o = types.SimpleNamespace(
        depth=True, heads={'hm': 2, 'wh': 2, 'params': 122}, iterations=False,
        PCA_SZ=63, knn_K=64, ball_radius=0.015, ball_radius2=0.04,
        sample_num_level1=512, sample_num_level2=128, INPUT_FEATURE_NUM=3, SAMPLE_NUM=1024,
        default_resolution=R, DECONV_DIMS=[256, 256, 256, 256], GCN_IN_DIM=[512, 256, 128],
        GCN_OUT_DIM=[256, 128, 64], IMG_DIMS=[256, 128, 64], graph_k=2, graph_layer_num=4)
    for k, v in over.items():
        setattr(o, k, v)
    return o


def surrogate_loss(res):
    """Scalar touching every model output (same definition as oracle/make_goldens.py)."""
    result, params, hand_list, other = res
    t = 0
    for h in ("left", "right"):
        t = t + result['verts3d'][h].pow(2).mean() + (result['verts2d'][h] / 384).pow(2).mean()
        t = t + params['scale'][h].pow(2).mean() + params['trans2d'][h].pow(2).mean() + params['root'][h].pow(2).mean()
        t = t + hand_list[0]['verts3d'][h].pow(2).mean()
    t = t + other['hms'].pow(2).mean() + other['mask'].pow(2).mean()
    for k in ('hm', 'wh', 'params'):
        t = t + other['ret'][k].pow(2).mean()
    return t


def pack_outputs(res, ind):
    result, params, hand_list, other = res
    o = {}
    for h in ("left", "right"):
        o["verts3d_" + h] = result['verts3d'][h]
        o["verts2d_" + h] = result['verts2d'][h]
        o["scale_" + h] = params['scale'][h]
        o["trans2d_" + h] = params['trans2d'][h]
        o["root_" + h] = params['root'][h]
        o["gcn_verts3d_" + h] = hand_list[0]['verts3d'][h]
        o["mano_list_verts3d_" + h] = other['verts3d_MANO_list'][h][0]
    B = ind.shape[0]
    p = other['ret']['params'].reshape(B, 122, -1)
    o["params_at_ind"] = torch.gather(p, 2, ind.unsqueeze(1).expand(B, 122, 2)).transpose(1, 2)
    o["hm"] = other['ret']['hm']
    o["wh_crop"] = other['ret']['wh'][:, :, 8:24, 8:24]
    for k in ("hms", "mask"):
        t = other[k]
        o[k + "_sum"] = t.double().sum().reshape(1)
        o[k + "_abs_sum"] = t.double().abs().sum().reshape(1)
        o[k + "_crop"] = t[:, :, 8:24, 8:24]
    return o


# tolerances: SURVEY.md Appendix C noise floor (fp32 vs fp64 of the oracle itself)
TOL_EVAL = {"verts3d": 1e-4, "scale": 1e-4, "trans2d": 1e-4, "root": 1e-4, "gcn_verts3d": 1e-4,
            "mano_list_verts3d": 1e-4, "params_at_ind": 1e-4, "hm": 1e-4}


def check_packed(got, exp, abs_tol=1e-4, rel_tol=1e-5, skip=()):
    """got: dict of tensors, exp: dict of numpy arrays. verts2d/hms/mask use relative tolerance."""
    bad = []
    for k, e in exp.items():
        if k not in got or k in skip:
            continue
        g = got[k].detach().cpu().double().numpy()
        e = e.astype(np.float64)
        d = np.abs(g - e).max()
        lim = abs_tol + rel_tol * np.abs(e).max()
        if k.startswith("verts2d") or k.startswith("hms") or k.startswith("mask"):
            lim = abs_tol + 5 * rel_tol * np.abs(e).max()      # pixel-scaled outputs: relative
        if not d <= lim:
            bad.append((k, d, lim))
    assert not bad, bad


def synthetic_model_outputs(B, R, seed):
    """Model-output-shaped tensors for the loss parity fixture (same generator as oracle/make_goldens.py)."""
    g = np.random.Generator(np.random.PCG64(seed))
    f = lambda *sh, sc=1.0: torch.from_numpy((g.standard_normal(sh) * sc).astype(np.float32))
    result = {'verts3d': {h: f(B, 778, 3, sc=0.05) for h in ('left', 'right')},
              'verts2d': {h: f(B, 778, 2, sc=30.0) + R / 2 for h in ('left', 'right')}}
    params = {'scale': {h: f(B, sc=0.3) for h in ('left', 'right')}, 'trans2d': {h: f(B, 2, sc=0.3) for h in ('left', 'right')},
              'root': {h: f(B, 3, sc=3.0) for h in ('left', 'right')}}
    hand = [{'verts3d': {h: f(B, 252, 3, sc=0.05) for h in ('left', 'right')},
             'verts2d': {h: f(B, 252, 2, sc=30.0) + R / 2 for h in ('left', 'right')}}]
    other = {'hms': f(B, 42, R // 4, R // 4, sc=0.3), 'mask': f(B, 2, R, R, sc=0.5),
             'ret': {'hm': f(B, 2, R // 4, R // 4) - 2.0, 'wh': f(B, 2, R // 4, R // 4), 'params': f(B, 122, R // 4, R // 4)}}
    return result, params, hand, other


def tree_to(obj, device):
    if torch.is_tensor(obj):
        return obj.to(device)
    if isinstance(obj, dict):
        return {k: tree_to(v, device) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(tree_to(v, device) for v in obj)
    return obj


def demo_fixture_inputs(device='cpu'):
    """BASELINE config 1: the network input of the reference's demo (demo.py:117-202) rebuilt from
    tests/golden/demo_H2O_000002_R256.npz -- uint8 BGR image -> ImageNet-normalised NCHW, uint16 millimetres -> metres --
    plus the clouds the reference's own depth2pcl produced.  -> (golden dict, batch dict)."""
    g = gold("demo_H2O_000002_R256")
    mean = np.array([0.485, 0.456, 0.406], np.float32).reshape(1, 1, 3)
    std = np.array([0.229, 0.224, 0.225], np.float32).reshape(1, 1, 3)
    pre = ((g["image_u8"].astype(np.float32) / 255. - mean) / std).astype(np.float32)
    b = {'input': torch.from_numpy(pre).permute(2, 0, 1).unsqueeze(0).contiguous(),
         'depth': torch.from_numpy(g["depth_mm_u16"].astype(np.float32) / 1000.).reshape(1, 1, 256, 256),
         'K_new': torch.from_numpy(g["K_img"].astype(np.float32)).reshape(1, 3, 3),
         'valid': torch.ones(1, 2), 'choose': torch.from_numpy(g["choose"]).unsqueeze(0), 'cloud': torch.from_numpy(g["cloud"]).unsqueeze(0)}
    return g, {k: v.to(device) for k, v in b.items()}


def demo_state_dict(template, g):
    """Generator weights + the mask-head bias shift the fixture was made with (oracle/make_demo_golden.py)."""
    from oracle import synth
    sd = synth.det_state_dict(template)
    sd['encoder.dp_decoder.final_layer.1.bias'] = sd['encoder.dp_decoder.final_layer.1.bias'] + torch.from_numpy(g["dp_bias_add"])
    return sd


def pack_demo(res):
    result, params, hand, other = res
    o = {}
    for h in ("left", "right"):
        o["verts3d_" + h] = result['verts3d'][h]
        o["verts2d_" + h] = result['verts2d'][h]
        o["scale_" + h] = params['scale'][h]
        o["trans2d_" + h] = params['trans2d'][h]
        o["root_" + h] = params['root'][h]
        o["gcn_verts3d_" + h] = hand[0]['verts3d'][h]
    o["hm"] = other['ret']['hm']
    for k in ("hms", "mask"):
        t = other[k]
        o[k + "_sum"] = t.double().sum().reshape(1)
        o[k + "_abs_sum"] = t.double().abs().sum().reshape(1)
        o[k + "_crop"] = t[:, :, 8:24, 8:24]
    return o


def build_c_client(out_dir):
    """Compile tests/c_abi/c_client.c (plain C, gcc) against include/pdfnet_hip.h and libpdfnet_hip.so -> path of the binary."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, "pdfnet_amd")
    exe = os.path.join(str(out_dir), "c_client")
    cmd = ["gcc", "-O1", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(root, "include"), "-I", "/opt/rocm/include", "-D__HIP_PLATFORM_AMD__",
           os.path.join(root, "tests", "c_abi", "c_client.c"), "-o", exe, "-L", lib, "-lpdfnet_hip", "-L", "/opt/rocm/lib", "-lamdhip64", "-lm",
           "-Wl,-rpath," + lib, "-Wl,-rpath,/opt/rocm/lib"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe
