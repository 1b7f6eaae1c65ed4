"""TEST INFRASTRUCTURE (oracle side) -- imports the *reference* PDFNet on CPU.

Only usable in the build container where /root/reference exists; never on the GPU box and never
from the product package.  It injects the stub modules the reference needs at import time
 without modifying the reference, and exposes

    ref_opt(R)            -- an `opt` namespace with the fields lib/opts.py:221-239,284-308 produce
    build_ref_model(R)    -- lib/models/networks/intaghand_model.py:49-67 `load_model_intag(opt)`
                             with the 256-vs-384 `img_size` patch (DualGraph.py:69-72)

Used by oracle/make_goldens.py (fixture generator) and tests that pin the restatement against the
reference when /root/reference is present.
"""
import os
import sys
import types
import importlib

import numpy as np
import torch

REF_ROOT = os.environ.get("PDFNET_REFERENCE", "/root/reference")


def reference_available():
    return os.path.isdir(os.path.join(REF_ROOT, "lib", "models", "networks"))


def _mod(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


_installed = False


def install_stubs():
    """Stub modules listed in SURVEY.md 8(c). Harness-side only."""
    global _installed
    if _installed:
        return
    _installed = True
    if not hasattr(np, "int"):
        np.int = int
    if not hasattr(np, "float"):
        np.float = float
    if not hasattr(np, "bool"):
        np.bool = bool

    # cv2: only resize (identity when sizes match) + no-op drawing/io
    def _resize(img, size, *a, **k):
        if tuple(img.shape[:2]) == (size[1], size[0]):
            return img
        # nearest resize fallback (masks)
        ys = (np.arange(size[1]) * img.shape[0] / size[1]).astype(np.int64)
        xs = (np.arange(size[0]) * img.shape[1] / size[0]).astype(np.int64)
        return img[ys][:, xs]

    def _noop(*a, **k):
        return None
    _mod("cv2", resize=_resize, imwrite=_noop, circle=_noop, imread=_noop, imshow=_noop,
         waitKey=_noop, line=_noop, putText=_noop, INTER_LINEAR=1, INTER_NEAREST=0,
         warpAffine=_noop, getAffineTransform=_noop, COLOR_BGR2RGB=4, cvtColor=_noop,
         FONT_HERSHEY_SIMPLEX=0, LINE_AA=16, rectangle=_noop, THRESH_BINARY=0, threshold=_noop,
         IMREAD_UNCHANGED=-1, IMREAD_ANYDEPTH=2, BORDER_CONSTANT=0, Rodrigues=_noop, flip=_noop)

    sys.path.insert(0, REF_ROOT)
    sys.path.insert(0, os.path.join(REF_ROOT, "lib"))

    # torchvision.models -> the reference's vendored twin lib/models/networks/resnet.py:76-122
    tv = _mod("torchvision")
    ref_resnet = importlib.import_module("lib.models.networks.resnet")
    tvm = _mod("torchvision.models", resnet18=ref_resnet.resnet18, resnet34=ref_resnet.resnet34,
               resnet50=ref_resnet.resnet50, resnet101=ref_resnet.resnet101,
               resnet152=ref_resnet.resnet152)
    tv.models = tvm

    class _Resize:
        def __init__(self, *a, **k):
            pass

        def __call__(self, x):
            return x
    tvt = _mod("torchvision.transforms", Resize=_Resize)
    tv.transforms = tvt

    # pytorch3d (render only)
    _mod("pytorch3d")
    _mod("pytorch3d.structures", Meshes=object)
    _mod("pytorch3d.renderer")
    _mod("pytorch3d.renderer.mesh")
    _mod("pytorch3d.renderer.mesh.textures", Textures=object)

    class _Bar:
        def __init__(self, *a, **k):
            self.suffix = ""

        def next(self):
            pass

        def finish(self):
            pass
    _mod("progress")
    _mod("progress.bar", Bar=_Bar)
    _mod("skimage")
    _mod("skimage.io")
    _mod("tkinter")
    _mod("tkinter.messagebox", NO="no")

    # chumpy so MANO_*.pkl unpickle (manolayer.py:108,141-144)
    class Ch:
        def __setstate__(self, state):
            self.__dict__.update(state)

        @property
        def r(self):
            return np.asarray(self.x)

        def __array__(self, dtype=None, copy=None):
            return np.asarray(self.r, dtype=dtype)

    class Select(Ch):
        @property
        def r(self):
            a = self.a.r if hasattr(self.a, "r") else np.asarray(self.a)
            return a.ravel()[self.idxs].reshape(self.preferred_shape)
    ch = _mod("chumpy", Ch=Ch)
    ch.ch = _mod("chumpy.ch", Ch=Ch)
    ch.reordering = _mod("chumpy.reordering", Select=Select)


def ref_opt(R=256, **over):
    """opt fields the model reads (lib/opts.py:221-239; heads at :291-295)."""
    o = types.SimpleNamespace(
        depth=True, heads={'hm': 2, 'wh': 2, 'params': 122}, iterations=False,
        PCA_SZ=63, knn_K=64, ball_radius=0.015, ball_radius2=0.04,
        sample_num_level1=512, sample_num_level2=128, INPUT_FEATURE_NUM=3, SAMPLE_NUM=1024,
        default_resolution=R, DECONV_DIMS=[256, 256, 256, 256], GCN_IN_DIM=[512, 256, 128],
        GCN_OUT_DIM=[256, 128, 64], IMG_DIMS=[256, 128, 64], graph_k=2, graph_layer_num=4,
        size_train=[R, R], local_rank=0, input_res=R)
    for k, v in over.items():
        setattr(o, k, v)
    return o


def build_ref_model(R=256, opt=None):
    install_stubs()
    from lib.models.networks.intaghand_model import load_model_intag
    opt = opt or ref_opt(R)
    model = load_model_intag(opt)
    # decoder hard-wired to 384 (intaghand_decoder.py:130; DualGraph.py:69-72)
    for i, layer in enumerate(model.decoder.dual_gcn.layers):
        layer.img_size = R // 32 * (2 ** i)
    return model


def ref_module(path):
    install_stubs()
    return importlib.import_module(path)
