"""TEST INFRASTRUCTURE -- fixture generator for BASELINE config 1 ("demo.py on one assets/H2O RGB-D pair, CPU-only
forward").  Runs ONLY in the build container: it reads the reference's own asset pair
/root/reference/assets/H2O/{color,depth}/000002.png, restates the preprocessing of demo.py:117-193 (cv2 is not installed,
so the PNGs are decoded with PIL and resampled here -- SURVEY.md 8(c): the golden therefore STARTS AT THE NETWORK INPUT),
pushes the result through the imported reference model exactly as demo.py:202 calls it
(`model(pre_img, None, None, depth_256, None, K_img, valid)`: centres from the heat-map, clouds from the depth map and the
predicted masks by the reference's numpy `depth2pcl` under np.random.seed) and writes

    tests/golden/demo_H2O_000002_R256.npz   inputs : image_u8 [256,256,3] BGR, depth_mm_u16 [256,256], K_img [3,3]
                                            captured: choose i64 [2,1024], cloud f32 [2,1024,3] (what depth2pcl produced)
                                            outputs: eval-mode result / paramsDict / hm / predicted centres / mask + hms digests
                                                     and the demo's decoded absolute joints (demo.py:213-232)

The reference's source never travels; the fixture is data (a 256x256 crop-resample of one image the reference ships, and
network outputs under the deterministic weight generator of oracle/synth.py).

    python oracle/make_demo_golden.py
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import ref_harness as rh          # noqa: E402
from oracle import synth                       # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
MEAN = np.array([0.485, 0.456, 0.406], np.float32).reshape(1, 1, 3)       # demo.py:108-111
STD = np.array([0.229, 0.224, 0.225], np.float32).reshape(1, 1, 3)
DP_BIAS_ADD = [0.37, 0.31]


def affine_from_points(src, dst):
    """cv2.getAffineTransform: the 2x3 matrix mapping three src points onto three dst points."""
    A = np.concatenate([src, np.ones((3, 1))], 1).astype(np.float64)
    return np.linalg.solve(A, dst.astype(np.float64)).T                     # [2,3]


def get_affine_transform(center, scale, out_size):
    """lib/utils/image.py:27-58 with rot = 0, shift = 0."""
    src_w, dst_w, dst_h = scale, out_size[0], out_size[1]
    src = np.zeros((3, 2), np.float32)
    dst = np.zeros((3, 2), np.float32)
    src[0] = center
    src[1] = center + np.array([0, src_w * -0.5], np.float32)
    dst[0] = [dst_w * 0.5, dst_h * 0.5]
    dst[1] = np.array([dst_w * 0.5, dst_h * 0.5], np.float32) + np.array([0, dst_w * -0.5], np.float32)
    third = lambda a, b: b + np.array([-(a - b)[1], (a - b)[0]], np.float32)       # get_3rd_point, image.py:80-83
    src[2] = third(src[0], src[1])
    dst[2] = third(dst[0], dst[1])
    return affine_from_points(src, dst), affine_from_points(dst, src)


def warp(img, inv, R, nearest):
    """cv2.warpAffine's geometry (inverse map of pixel centres, constant 0 border); bilinear or nearest."""
    ys, xs = np.mgrid[0:R, 0:R].astype(np.float64)
    sx = inv[0, 0] * xs + inv[0, 1] * ys + inv[0, 2]
    sy = inv[1, 0] * xs + inv[1, 1] * ys + inv[1, 2]
    H, W = img.shape[:2]
    im = img.astype(np.float64)
    if im.ndim == 2:
        im = im[..., None]
    if nearest:
        xi, yi = np.rint(sx).astype(np.int64), np.rint(sy).astype(np.int64)
        ok = (xi >= 0) & (xi < W) & (yi >= 0) & (yi < H)
        out = np.where(ok[..., None], im[np.clip(yi, 0, H - 1), np.clip(xi, 0, W - 1)], 0.0)
    else:
        x0, y0 = np.floor(sx).astype(np.int64), np.floor(sy).astype(np.int64)
        fx, fy = (sx - x0)[..., None], (sy - y0)[..., None]

        def px(yy, xx):
            ok = (xx >= 0) & (xx < W) & (yy >= 0) & (yy < H)
            return np.where(ok[..., None], im[np.clip(yy, 0, H - 1), np.clip(xx, 0, W - 1)], 0.0)
        out = (px(y0, x0) * (1 - fx) + px(y0, x0 + 1) * fx) * (1 - fy) + (px(y0 + 1, x0) * (1 - fx) + px(y0 + 1, x0 + 1) * fx) * fy
    return out if img.ndim == 3 else out[..., 0]


def main(R=256, name="000002"):
    from PIL import Image
    rh.install_stubs()
    col = np.asarray(Image.open(os.path.join(rh.REF_ROOT, "assets/H2O/color/%s.png" % name)).convert("RGB"))[..., ::-1]   # cv2.imread: BGR
    dep = np.asarray(Image.open(os.path.join(rh.REF_ROOT, "assets/H2O/depth/%s.png" % name))).astype(np.float64)          # 16-bit, millimetres
    assert col.shape == (720, 1280, 3) and dep.shape == (720, 1280)
    K = np.array([[636.6593017578125, 0, 635.283881879317], [0, 636.251953125, 366.8740353496978], [0, 0, 1]])           # demo.py:132-133
    K[0, 2], K[1, 2] = K[1, 2], K[0, 2]                                                                                     # demo.py:135-137 (as written)
    h, w = col.shape[:2]
    trans, inv = get_affine_transform(np.array([w / 2., h / 2.], np.float32), max(h, w) * 1., [R, R])
    K_img = K.copy()                                                                                                        # demo.py:144-148
    K_img[0, 0], K_img[1, 1] = K[0, 0] * trans[0, 0], K[1, 1] * trans[1, 1]
    K_img[0, 2] = K[0, 2] * trans[0, 0] + trans[0, 2]
    K_img[1, 2] = K[1, 2] * trans[1, 1] + trans[1, 2]
    image_u8 = np.clip(np.rint(warp(col, inv, R, nearest=False)), 0, 255).astype(np.uint8)                                 # demo.py:179-181
    depth_mm = np.clip(np.rint(warp(dep, inv, R, nearest=True)), 0, 65535).astype(np.uint16)                               # demo.py:193-195
    # ---- network input, exactly as the tests rebuild it from the fixture
    pre = ((image_u8.astype(np.float32) / 255. - MEAN) / STD).astype(np.float32)                                            # demo.py:325-326
    pre_img = torch.from_numpy(pre).permute(2, 0, 1).unsqueeze(0)
    depth_256 = depth_mm.astype(np.float32) / 1000.
    valid = np.array([[1, 1]])

    model = rh.build_ref_model(R)
    sd = synth.det_state_dict(model.state_dict())
    # The generator weights give mask logits of 0.09..0.22 on this image: no pixel passes the reference's `mask > 0.5`
    # (intaghand_encoder.py:372) and both clouds would be all-zero.  Shift the mask head's bias so that about half of the
    # pixels pass -- the front end then sees real depth: z-window, > 1024 candidates, random subset.  The tests apply the
    # same shift (stored in the fixture).
    sd['encoder.dp_decoder.final_layer.1.bias'] = sd['encoder.dp_decoder.final_layer.1.bias'] + torch.tensor(DP_BIAS_ADD)
    model.load_state_dict(sd)
    model.eval()
    E = rh.ref_module("lib.models.networks.intaghand_encoder")
    captured = {}
    real = E.depth2pcl

    def spy(*a, **k):
        ch, cl = real(*a, **k)
        captured['choose'], captured['cloud'] = ch.copy(), cl.copy()
        return ch, cl
    E.depth2pcl = spy
    np.random.seed(317)                                                                                                     # main.py:37-44 seed
    with torch.no_grad():
        result, params, hand, other = model(pre_img, None, None, torch.from_numpy(depth_256), None, torch.from_numpy(K_img), valid)
    E.depth2pcl = real
    out = {"image_u8": image_u8, "depth_mm_u16": depth_mm, "K_img": K_img.astype(np.float64), "dp_bias_add": np.array(DP_BIAS_ADD, np.float32),
           "choose": captured['choose'].astype(np.int64), "cloud": captured['cloud'].astype(np.float32)}
    for hnd in ("left", "right"):
        out["verts3d_" + hnd] = result['verts3d'][hnd].numpy()
        out["verts2d_" + hnd] = result['verts2d'][hnd].numpy()
        out["scale_" + hnd] = params['scale'][hnd].numpy()
        out["trans2d_" + hnd] = params['trans2d'][hnd].numpy()
        out["root_" + hnd] = params['root'][hnd].numpy()
        out["gcn_verts3d_" + hnd] = hand[0]['verts3d'][hnd].numpy()
    hm = other['ret']['hm']
    out["hm"] = hm.numpy()
    nms = E._nms(hm.clone(), 5)                                                                                             # the encoder's own pick (:750-758)
    out["pred_ind"] = torch.cat((E._topk(nms[:, :1], 1)[1], E._topk(nms[:, 1:], 1)[1]), 1).numpy()
    # the demo's pick on the sigmoid map and its decode (demo.py:203-232)
    from lib.models.utils import _sigmoid
    chm = E._nms(_sigmoid(hm.clone()), 5)
    il, ir = E._topk(chm[:, :1], 1)[1], E._topk(chm[:, 1:], 1)[1]
    out["demo_ind"] = torch.cat((il, ir), 1).numpy()
    mask = other['mask']
    out["mask_pos_count"] = np.array([(mask[0, 0] > 0.5).sum().item(), (mask[0, 1] > 0.5).sum().item()])
    for k in ("hms", "mask"):
        t = other[k]
        out[k + "_sum"] = t.double().sum().reshape(1).numpy()
        out[k + "_abs_sum"] = t.double().abs().sum().reshape(1).numpy()
        out[k + "_crop"] = t[:, :, 8:24, 8:24].numpy()
    p = os.path.join(OUT, "demo_H2O_%s_R%d.npz" % (name, R))
    np.savez_compressed(p, **out)
    uniq = [len(np.unique(captured['choose'][i])) for i in range(2)]
    print("wrote %s (%.1f KB); mask>0.5 pixels (right,left) %s; unique cloud points (left,right) %s; pred_ind %s"
          % (p, os.path.getsize(p) / 1024, out["mask_pos_count"].tolist(), uniq, out["pred_ind"].tolist()))


if __name__ == "__main__":
    main()
