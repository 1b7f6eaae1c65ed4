#!/usr/bin/env python3
"""MANO_{LEFT,RIGHT}.pkl -> mano_constants.npz: the constants the hot path needs, extracted at the USER'S site.

    python tools/convert_mano.py --left .../MANO_LEFT.pkl --right .../MANO_RIGHT.pkl --out mano_constants.npz [--no-fix-shape]

The MPI-licensed pickles are never committed and neither is the output (`mano_constants.npz` is git-ignored); this script is what
turns them into the plain arrays `pdfnet_amd.utils.load_mano_constants` hands to `CtdetLoss(opt, consts)` and `F.mano_lbs(consts, ...)`.
The pickles hold chumpy objects; chumpy is not needed: two tiny stand-in classes with the attribute layout the pickles use are
registered for the duration of the load (the same ones the oracle harness uses, oracle/ref_harness.py).

What is written, per side s in (left, right), following the reference field by field:
  v_template_s [778,3], shapedirs_s [778,3,10] (chumpy Select of 10 of 20 columns), posedirs_s [778,3,135], J_regressor_s [16,778] dense,
  weights_s [778,16], hands_components_s [45,45], hands_mean_s [45]                           lib/models/networks/manolayer.py:100-160
  faces_s [1538,3] int64                                                                       manolayer.py:146  (`manoData['f']`)
  full_regressor_s [21,778] = rows of J_regressor + five one-hot tip rows (745, 317, 444, 556, 673), reordered by new_order
                                                                                               lib/models/hand3d/Mano_model.py:309-323
and once: kintree_parents [16], new_order [21], fix_shape (0/1): the sign flip of left.shapedirs[:,0,:] the loss module applies when the
two sides' first shape directions coincide (lib/datasets/interhand.py:120-123, applied at lib/trains/simplified.py:52) -- ON by default,
because CtdetLoss is what consumes these constants; pass --no-fix-shape for the dataset's per-item layers (interhand.py:460-461)."""
import argparse
import hashlib
import pickle
import sys
import types

import numpy as np

NEW_ORDER = [0, 13, 14, 15, 16, 1, 2, 3, 17, 4, 5, 6, 18, 10, 11, 12, 19, 7, 8, 9, 20]      # manolayer.py:110-115, Mano_model.py:319-322
TIPS = (745, 317, 444, 556, 673)                                                           # Mano_model.py:312-316


class _Ch:
    def __setstate__(self, state):
        self.__dict__.update(state)

    @property
    def r(self):
        return np.asarray(self.x)

    def __array__(self, dtype=None, copy=None):
        return np.asarray(self.r, dtype=dtype)


class _Select(_Ch):
    @property
    def r(self):
        a = self.a.r if hasattr(self.a, "r") else np.asarray(self.a)
        return a.ravel()[self.idxs].reshape(self.preferred_shape)


def _load_pickle(path):
    saved = {k: sys.modules.get(k) for k in ("chumpy", "chumpy.ch", "chumpy.reordering")}
    if saved["chumpy"] is None:                                   # a real chumpy, when installed, is used as is
        ch = types.ModuleType("chumpy")
        ch.Ch = _Ch
        ch.ch = types.ModuleType("chumpy.ch")
        ch.ch.Ch = _Ch
        ch.reordering = types.ModuleType("chumpy.reordering")
        ch.reordering.Select = _Select
        sys.modules.update({"chumpy": ch, "chumpy.ch": ch.ch, "chumpy.reordering": ch.reordering})
    try:
        with open(path, "rb") as f:
            return pickle.load(f, encoding="latin1")
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)


def _arr(v, dtype=np.float32):
    return np.ascontiguousarray(np.asarray(v.r if hasattr(v, "r") else v), dtype=dtype)


def extract_side(path):
    d = _load_pickle(path)
    out = {
        "v_template": _arr(d["v_template"]), "shapedirs": _arr(d["shapedirs"]), "posedirs": _arr(d["posedirs"]),
        "J_regressor": np.ascontiguousarray(d["J_regressor"].toarray(), dtype=np.float32), "weights": _arr(d["weights"]),
        "hands_components": _arr(d["hands_components"]), "hands_mean": _arr(d["hands_mean"]),
        "faces": np.ascontiguousarray(np.asarray(d["f"]), dtype=np.int64),
        "kintree_parents": np.asarray([-1] + [int(d["kintree_table"][0, i]) for i in range(1, 16)], dtype=np.int64),
    }
    assert out["v_template"].shape == (778, 3) and out["shapedirs"].shape == (778, 3, 10) and out["posedirs"].shape == (778, 3, 135)
    assert out["J_regressor"].shape == (16, 778) and out["weights"].shape == (778, 16) and out["faces"].shape == (1538, 3)
    tip = np.zeros((5, 778), np.float32)
    for r, v in enumerate(TIPS):
        tip[r, v] = 1.0
    out["full_regressor"] = np.ascontiguousarray(np.concatenate([out["J_regressor"], tip], 0)[NEW_ORDER])
    return out


def convert(left_pkl, right_pkl, fix_shape=True):
    sides = {"left": extract_side(left_pkl), "right": extract_side(right_pkl)}
    assert np.array_equal(sides["left"]["kintree_parents"], sides["right"]["kintree_parents"])
    flipped = 0
    if fix_shape and np.abs(sides["left"]["shapedirs"][:, 0, :] - sides["right"]["shapedirs"][:, 0, :]).sum() < 1:       # interhand.py:120-123
        sides["left"]["shapedirs"][:, 0, :] *= -1
        flipped = 1
    z = {"kintree_parents": sides["left"]["kintree_parents"], "new_order": np.asarray(NEW_ORDER, np.int64), "fix_shape": np.asarray(flipped, np.int64)}
    for s, d in sides.items():
        for k, v in d.items():
            if k != "kintree_parents":
                z["%s_%s" % (k, s)] = v
    return z


def digests(z):
    return {k: hashlib.sha256(np.ascontiguousarray(v).tobytes()).hexdigest() for k, v in sorted(z.items())}


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--left", required=True)
    ap.add_argument("--right", required=True)
    ap.add_argument("--out", default="mano_constants.npz")
    ap.add_argument("--no-fix-shape", action="store_true")
    ap.add_argument("--print-sha256", action="store_true", help="print the SHA-256 of every array (what tests/golden/mano_constants.sha256 pins)")
    a = ap.parse_args()
    z = convert(a.left, a.right, fix_shape=not a.no_fix_shape)
    np.savez(a.out, **z)
    print("wrote %s: %d arrays, fix_shape=%d" % (a.out, len(z), int(z["fix_shape"])))
    if a.print_sha256:
        for k, h in digests(z).items():
            print("%s  %s" % (h, k))


if __name__ == "__main__":
    main()
