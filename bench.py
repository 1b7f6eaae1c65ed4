#!/usr/bin/env python3
"""Headline benchmark: train-step images/s @ 256x256 RGB-D, B=32 per GPU, fp32 (BASELINE.json configs[2]).

    python bench.py --gpus 1 --steps 50 --warmup 10        (the defaults: SURVEY 8(d) asks for >= 50 steps after >= 10 warm-up)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One step = model forward + CtdetLoss + backward + gradient all-reduce (N>1) + Adam, on a synthetic batch that
is already resident in HBM.  Rank 0 prints ONE JSON line (contract in the task description) that also carries
  roofline     : fp32-MFMA roofline of the dominant kernel family (the implicit-GEMM kernels), measured live
                 with events on the launch stream in an extra instrumented step after the timed region;
  cpu_baseline : the CPU oracle (oracle/pdfnet_cpu.py) timed on this box's host cores on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_FP32_MFMA_TFLOPS = 157.3          # /opt/skills/guides/MI355X_MICROARCH.md:42
PEAK_BF16_MFMA_TFLOPS = 2500.0         # /opt/skills/guides/MI355X_MICROARCH.md:43 (dense)
ALGO_GFLOP_PER_IMG_STEP_DENSE = 358.0  # 119.46 GF/img forward x3 (reference formulation)


def make_opt(R):
    import types
    return types.SimpleNamespace(
        depth=True, heads={'hm': 2, 'wh': 2, 'params': 122}, iterations=False, PCA_SZ=63, knn_K=64,
        ball_radius=0.015, ball_radius2=0.04, sample_num_level1=512, sample_num_level2=128, INPUT_FEATURE_NUM=3,
        SAMPLE_NUM=1024, default_resolution=R, DECONV_DIMS=[256, 256, 256, 256], GCN_IN_DIM=[512, 256, 128],
        GCN_OUT_DIM=[256, 128, 64], IMG_DIMS=[256, 128, 64], graph_k=2, graph_layer_num=4,
        size_train=[R, R], down_ratio=4, center_weight=200.0, reproj_weight=1.0, bone_dir_weight=200.0)


def _with_explicit_form(lib, name):
    """-> [name, name + '_x'] when the library has the explicit-options form of the entry point (include/pdfnet_hip.h PdfCallOpts):
    same positional arguments, the options structure appended -- the argument positions the profilers read are unchanged."""
    names = [name, name + '_x'] if (name + '_x') in lib.protos else [name]
    if name.startswith('pdf_mesh_level'):                    # the x3 and bf16 builds of the fused mesh decoder: same call, same contraction
        names += [name + s for s in ('_x3', '_bf16') if (name + s) in lib.protos]
    return names


class GemmProfiler:
    """Wraps the GEMM-family C-ABI entry points with start/stop events on the current stream and counts the
    algorithmic FLOPs (2*M*N*K of the contraction each call performs) from the call's own arguments."""
    NAMES = ('pdf_linear_fwd', 'pdf_linear_bwd_data', 'pdf_linear_bwd_weight', 'pdf_linear_fwd_pair', 'pdf_linear_bwd_data_pair',
             'pdf_linear_bwd_weight_pair', 'pdf_conv2d_fwd', 'pdf_conv2d_bwd_data', 'pdf_conv2d_bwd_weight',
             'pdf_deconv2d_fwd', 'pdf_deconv2d_bwd_data', 'pdf_deconv2d_bwd_weight', 'pdf_mesh_level_fwd', 'pdf_mesh_level_bwd')
    # position of (M, N, K) in the argument list and the number of GEMMs per call, include/pdfnet_hip.h
    LINEAR = {'pdf_linear_fwd': (4, 1), 'pdf_linear_bwd_data': (3, 1), 'pdf_linear_bwd_weight': (6, 1),
              'pdf_linear_fwd_pair': (6, 2), 'pdf_linear_bwd_data_pair': (4, 2), 'pdf_linear_bwd_weight_pair': (8, 2)}

    def __init__(self):
        from pdfnet_amd import hip
        self.lib = hip.lib()
        self.records = []
        self.saved = {}

    @staticmethod
    def mesh_level_flops(level, B):
        """Forward contraction FLOPs of one DualGraphLayer (csrc/meshdec.hip): its 28 Linear layers per hand pair + the two attentions."""
        V, cin, c = 63 << level, 512 >> level, 256 >> level
        R = 2 * B * V
        lin = 2 * R * c * (2 * cin + 2 * c + cin) + 3 * 2 * R * c * (2 * c + 2 * c + c) + 2 * 6 * 2 * R * c * c
        att = 2 * 2 * B * 4 * 4 * V * V * (c // 4)
        return float(lin), float(att)

    @staticmethod
    def flops(name, a):
        if name.startswith('pdf_mesh_level'):
            lv = a[0]._obj                                      # the PdfMeshLevel structure behind byref()
            lin, att = GemmProfiler.mesh_level_flops(lv.level, lv.B)
            # backward: data gradients + weight gradients of every Linear (the call issues both), ~2.5x the attention's forward work
            return lin + att if name.endswith('fwd') else 2.0 * lin + 2.5 * att
        if name in GemmProfiler.LINEAR:
            off, n = GemmProfiler.LINEAR[name]
            return 2.0 * n * a[off] * a[off + 1] * a[off + 2]
        # conv family: (..., N, H, W, Cin, ld, Cout, KH, KW, stride, pad, OH, OW, ...)
        off = {'pdf_conv2d_fwd': 4, 'pdf_conv2d_bwd_data': 3, 'pdf_conv2d_bwd_weight': 6,
               'pdf_deconv2d_fwd': 4, 'pdf_deconv2d_bwd_data': 3, 'pdf_deconv2d_bwd_weight': 5}[name]
        N, H, W, Cin, _, Cout, KH, KW, stride, pad, OH, OW = a[off:off + 12]
        if name.startswith('pdf_conv2d'):
            return 2.0 * N * OH * OW * Cout * Cin * KH * KW
        return 2.0 * N * H * W * Cin * Cout * KH * KW          # transposed conv: per input pixel

    def __enter__(self):
        for base in self.NAMES:
            for n in _with_explicit_form(self.lib, base):         # the plain entry point and its `_x` form (explicit PdfCallOpts last)
                fn = getattr(self.lib, n)                           # materialises the bound wrapper
                self.saved[n] = fn

                def wrapped(*a, _fn=fn, _n=base):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    n0 = self.lib.pdf_debug_igemm_launches()
                    e0.record()
                    r = _fn(*a)
                    e1.record()
                    weight = 'weight' in _n
                    mesh = _n.startswith('pdf_mesh_level')
                    tile = -2 if mesh else -1 if weight else self.lib.pdf_debug_last_tile()
                    launches = (3 if _n.endswith('fwd') else 8) if mesh else 1 if weight else self.lib.pdf_debug_igemm_launches() - n0
                    ints = (a[0]._obj.level, a[0]._obj.B) if mesh else tuple(v for v in a if isinstance(v, int) and abs(v) < (1 << 31))
                    self.records.append((_n, self.flops(_n, a), e0, e1, ints, tile, launches))
                    return r
                setattr(self.lib, n, wrapped)
        return self

    def __exit__(self, *exc):
        for n, fn in self.saved.items():
            setattr(self.lib, n, fn)

    def summary(self):
        torch.cuda.synchronize()
        per = {}
        for n, fl, e0, e1, _, _t, _l in self.records:
            d = per.setdefault(n, [0, 0.0, 0.0])
            d[0] += 1
            d[1] += fl
            d[2] += e0.elapsed_time(e1) * 1e-3
        return per

    def by_tile(self):
        """tile code (pdf_debug_last_tile; -1 = weight-gradient kernels) -> [kernel launches, flops, seconds]."""
        torch.cuda.synchronize()
        out = {}
        for n, fl, e0, e1, _, t, launches in self.records:
            d = out.setdefault(t, [0, 0.0, 0.0])
            d[0] += launches
            d[1] += fl
            d[2] += e0.elapsed_time(e1) * 1e-3
        return out

    def by_shape(self):
        """(entry point, integer arguments) -> [calls, flops, seconds], largest time first."""
        torch.cuda.synchronize()
        sh = {}
        for n, fl, e0, e1, ints, t, _l in self.records:
            d = sh.setdefault((n, ints + (('tile', t),)), [0, 0.0, 0.0])
            d[0] += 1
            d[1] += fl
            d[2] += e0.elapsed_time(e1) * 1e-3
        return sorted(sh.items(), key=lambda kv: -kv[1][2])


class KernelTimer:
    """The library's own per-kernel timing (pdf_debug_kernel_timing, csrc/gemm.hip KTimer): every GEMM-family kernel launch is
    bracketed by two HIP events on the stream it is launched on and recorded under its symbol name with the algorithmic FLOPs
    and bytes (every operand once) of that launch -- one record per kernel launch, not per entry-point call."""

    def __init__(self):
        from pdfnet_amd import hip
        self.lib = hip.lib()

    def __enter__(self):
        from pdfnet_amd import taped
        taped.SUSPEND = True                                 # (a replayed tape calls the library past the profilers' wrappers: they would miss the trunk)
        self.lib.pdf_debug_kernel_timing(1)
        return self

    def __exit__(self, *exc):
        from pdfnet_amd import taped
        taped.SUSPEND = False
        self.lib.pdf_debug_kernel_timing(0)

    def by_symbol(self):
        """kernel symbol -> [launches, flops, bytes, seconds]"""
        import ctypes
        torch.cuda.synchronize()
        out = {}
        name = ctypes.create_string_buffer(128)
        fl, by, ms = ctypes.c_double(), ctypes.c_double(), ctypes.c_float()
        for i in range(self.lib.pdf_debug_kernel_record_count()):
            self.lib.pdf_debug_kernel_record(i, name, 128, ctypes.byref(fl), ctypes.byref(by), ctypes.byref(ms))
            d = out.setdefault(name.value.decode(), [0, 0.0, 0.0, 0.0])
            d[0] += 1
            d[1] += fl.value
            d[2] += by.value
            d[3] += ms.value * 1e-3
        return out


def symbol_roofline(sym, peak, traffic_by_symbol=None, traffic_src=None):
    """-> (`roofline` fields of the symbol with the most time, per-symbol table)."""
    table = {k: {"launches": v[0], "gflop": round(v[1] / 1e9, 1), "ms": round(v[3] * 1e3, 3), "tflops": round(v[1] / max(v[3], 1e-9) / 1e12, 1),
                 "algorithmic_MB_per_launch": round(v[2] / max(v[0], 1) / 1e6, 1)} for k, v in sorted(sym.items(), key=lambda kv: -kv[1][3])}
    for k, row in table.items():
        if k.startswith('x3gemm'):                           # fp32-equivalent FLOPs executed as six bf16 MFMAs each: the pipe's own rate and peak beside them
            row["arithmetic"] = "x3: six bf16 MFMAs per fp32 product"
            row["bf16_mfma_tflops"] = round(6.0 * row["tflops"], 1)
            row["frac_of_bf16_mfma_peak"] = round(6.0 * row["tflops"] / PEAK_BF16_MFMA_TFLOPS, 4)
    mfma = {k: v for k, v in sym.items() if v[1] > 0}
    name, dom = max(mfma.items(), key=lambda kv: kv[1][3]) if mfma else ("none", [0, 0.0, 0.0, 1e-9])
    tr = (traffic_by_symbol or {}).get(name)
    head = {"kernel": name, "achieved": round(dom[1] / dom[3] / 1e12, 2), "frac": round(dom[1] / dom[3] / 1e12 / peak, 4),
            "launches_per_step": dom[0], "ms_per_step": round(dom[3] * 1e3, 3),
            "algorithmic_gflop_per_launch": round(dom[1] / max(dom[0], 1) / 1e9, 2), "avg_launch_ms": round(dom[3] / max(dom[0], 1) * 1e3, 4),
            "algorithmic_bytes_per_launch": round(dom[2] / max(dom[0], 1)), "traffic": tr, "traffic_source": traffic_src}
    if name.startswith('x3gemm'):
        # an x3 kernel runs on the bf16 matrix pipe (six MFMAs per fp32 product): its roofline is that pipe's -- achieved = the bf16 FLOPs it executes,
        # peak = 2.5 PFLOP/s; the fp32-equivalent rate stays beside it
        head["achieved_fp32_equivalent"] = head["achieved"]
        head["achieved"] = round(6.0 * dom[1] / dom[3] / 1e12, 2)
        head["peak"] = PEAK_BF16_MFMA_TFLOPS
        head["frac"] = round(head["achieved"] / PEAK_BF16_MFMA_TFLOPS, 4)
        head["arithmetic"] = "x3: six bf16 MFMAs per fp32 product (fp32-exact to ~2^-24), bf16 matrix pipe"
    # the kernel TEMPLATE with the most time when its instantiations are summed (two boolean instantiations of one
    # igemm_nt<64, 64, ...> loop were the largest pool of the step while `kernel` named a smaller single symbol)
    fam = {}
    for k, v in mfma.items():
        base = k.split('<')[0] + ('<' + ', '.join(k.split('<', 1)[1].split(', ')[:2]).rstrip('>') + ', ...>' if '<' in k else '')
        f = fam.setdefault(base, [0, 0.0, 0.0, 0.0])
        for i in range(4):
            f[i] += v[i]
    if fam:
        fname, fv = max(fam.items(), key=lambda kv: kv[1][3])
        head["largest_template"] = {"template": fname, "launches_per_step": fv[0], "ms_per_step": round(fv[3] * 1e3, 3),
                                    "achieved": round(fv[1] / max(fv[3], 1e-9) / 1e12, 2), "frac": round(fv[1] / max(fv[3], 1e-9) / 1e12 / peak, 4)}
    return head, table


def _isnull(v):
    return v is None or getattr(v, 'value', 1) in (None, 0)


class HbmProfiler:
    """The HBM-bound kernel family of the path (kNN / ball-query / grouping, gathers, set-abstraction tail,
    BatchNorm, L2Norm, bilinear x2, Adam; plus FPS, timed on its own because the live path never calls it): events on the
    launch stream around each C-ABI call, ALGORITHMIC bytes (every operand read / written once) from the call's own arguments."""
    PEAK_GBS = 8000.0                                    # /opt/skills/guides/MI355X_MICROARCH.md:36 (spec; 6.29 TB/s measured copy)

    @staticmethod
    def nbytes(name, a):
        f = 4
        if name == 'pdf_knn_ball_group':                 # pts, ldp, C, Bc, N, S, K, r2, idx, grouped, ldg
            _, ldp, C, Bc, N, S, K, _, _, grouped, ldg = a[:11]
            return Bc * (N * 3 * f + S * K * 4 + (0 if _isnull(grouped) else S * K * ldg * f + N * C * f))
        if name == 'pdf_gather_sub_fwd':                 # u, ldu, v, ldv, idx, Bc, N, S, K, C, y, ldy
            Bc, N, S, K, C = a[5:10]
            return Bc * (S * K * C * f + S * K * 4 + N * C * f + S * C * f)
        if name == 'pdf_gather_sub_bwd':                 # dy, lddy, idx, du, ldu, dv, ldv, Bc, N, S, K, C
            Bc, N, S, K, C = a[7:12]
            return Bc * (S * K * C * f + S * K * 4 + N * C * f + S * C * f)
        if name == 'pdf_gather_sub_bwd_sorted':          # dy, lddy, start, list, du, ldu, dv, ldv, Bc, N, S, K, C
            Bc, N, S, K, C = a[8:13]
            return Bc * (S * K * C * f + S * K * 4 + N * C * f + S * C * f)
        if name == 'pdf_bn_relu_maxk_fwd':               # y, ldy, C, R, K, ..., training at 11
            C, R, K = a[2:5]
            return R * K * C * f * (2 if a[11] else 1) + R * C * 8
        if name == 'pdf_bn_relu_maxk_bwd':               # dout, lddo, arg, y, ldy, mean, rstd, gamma, scale, shift, C, R, K
            C, R, K = a[10:13]
            return 2 * R * K * C * f + 3 * R * C * f
        if name == 'pdf_bn_train_fwd':                   # x, ldx, C, R, ..., res at 10
            C, R = a[2], a[3]
            return R * C * f * (3 + (0 if _isnull(a[10]) else 1))
        if name == 'pdf_bn_train_bwd':                   # dy, lddy, y, ldy, relu, x, ldx, ..., C at 12, R at 13, dx, lddx, dres
            C, R = a[12], a[13]
            return R * C * f * (5 + (0 if _isnull(a[16]) else 3))
        if name == 'pdf_gather_rows':                    # feat, ldf, C, HW, ind, stride, B, M, R, shift, out, ldo
            C, B, M, ldo = a[2], a[6], a[7], a[11]
            return B * M * (C * f + ldo * f + 8)
        if name == 'pdf_scatter_rows_add':
            ldo, C, B, M = a[1], a[2], a[6], a[7]
            return B * M * (2 * C * f + ldo * f + 8)
        if name in ('pdf_l2norm_fwd',):                  # x, ldx, C, R
            return a[2] * a[3] * f * 2
        if name in ('pdf_l2norm_bwd',):                  # dy, lddy, x, ldx, C, R
            return a[4] * a[5] * f * 3
        if name in ('pdf_upsample2x_fwd', 'pdf_upsample2x_bwd'):     # x, N, H, W, C
            return a[1] * a[2] * a[3] * a[4] * f * 5
        if name == 'pdf_adam_step':                      # p, g, m, v, n
            return a[4] * f * 7
        return 0

    NAMES = ('pdf_knn_ball_group', 'pdf_gather_sub_fwd', 'pdf_gather_sub_bwd', 'pdf_gather_sub_bwd_sorted', 'pdf_bn_relu_maxk_fwd', 'pdf_bn_relu_maxk_bwd',
             'pdf_bn_train_fwd', 'pdf_bn_train_bwd', 'pdf_gather_rows', 'pdf_scatter_rows_add', 'pdf_l2norm_fwd', 'pdf_l2norm_bwd',
             'pdf_upsample2x_fwd', 'pdf_upsample2x_bwd', 'pdf_adam_step')

    def __init__(self):
        from pdfnet_amd import hip
        self.lib = hip.lib()
        self.records, self.saved = [], {}
        self.knn_pairs = 0.0

    def __enter__(self):
        for base in self.NAMES:
            for n in _with_explicit_form(self.lib, base):
                fn = getattr(self.lib, n)
                self.saved[n] = fn

                def wrapped(*a, _fn=fn, _n=base):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    r = _fn(*a)
                    e1.record()
                    self.records.append((_n, float(self.nbytes(_n, a)), e0, e1, tuple(v for v in a if isinstance(v, int) and 0 < v < (1 << 32))))
                    if _n == 'pdf_knn_ball_group':           # pts, ldp, C, Bc, N, S, K, ...: Bc * S * N candidate (centroid, point) pairs
                        self.knn_pairs += float(a[3]) * a[4] * a[5]
                    return r
                setattr(self.lib, n, wrapped)
        return self

    def __exit__(self, *exc):
        for n, fn in self.saved.items():
            setattr(self.lib, n, fn)

    def summary(self):
        torch.cuda.synchronize()
        per = {}
        for n, b, e0, e1, _ in self.records:
            d = per.setdefault(n, [0, 0.0, 0.0])
            d[0] += 1
            d[1] += b
            d[2] += e0.elapsed_time(e1) * 1e-3
        return per

    def by_shape(self):
        torch.cuda.synchronize()
        per = {}
        for n, b, e0, e1, ints in self.records:
            d = per.setdefault((n, ints), [0, 0.0, 0.0])
            d[0] += 1
            d[1] += b
            d[2] += e0.elapsed_time(e1) * 1e-3
        return sorted(per.items(), key=lambda kv: -kv[1][2])


def fps_hbm(dev, Bc=64, N=4096, S=1024):
    """Farthest point sampling (pdf_fps; reference helper interhand.py:147-178) on its own: the live path never calls it
    (SURVEY 0.1).  One block per cloud keeps points and running distances in registers: 2 x N x 12 B read + S x 4 B written
    per cloud are ALL its HBM bytes, so its time is S dependent arg-max picks -- latency, not bandwidth (stated, not hidden)."""
    from pdfnet_amd import functional as F
    x = torch.rand(Bc, N, 3, device=dev)
    for _ in range(2):
        F.fps(x, S)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        F.fps(x, S)
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) * 1e-3 / 3
    nb = Bc * (N * 12 + S * 4)
    return {"clouds": Bc, "points": N, "picks": S, "ms": round(t * 1e3, 3), "us_per_pick": round(t / S * 1e6, 3),
            "algorithmic_MB": round(nb / 1e6, 2), "achieved_GBs": round(nb / t / 1e9, 2), "bound": "latency (S dependent arg-max rounds per cloud)",
            "kernel": "fps_wave_kernel (one wave per cloud, 16 points per lane, DPP arg-max)" if N <= 1024 else "fps_kernel (one 1,024-thread block per cloud)"}


def _flush_c_stdout():
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass


def pmc_traffic():
    """-> ({kernel symbol: HBM bytes per launch}, bytes per launch of the whole GEMM family, source tag).  PMC counters cannot be read
    from inside this process: the figures come from the last committed profile (tools/profile_step.sh -> profiles/*_pmc_traffic.json,
    separate --pmc FETCH_SIZE / WRITE_SIZE passes, FETCH_SIZE x2 on gfx950).  The profile records the sha256 of csrc/gemm.hip it
    was taken with; if the kernel source has changed since, the numbers are withheld (null) instead of going stale."""
    import glob
    import hashlib
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
    if not files:
        return None, None, "no PMC profile committed"
    p = files[-1]
    try:
        d = json.load(open(p))
        h = hashlib.sha256()
        for f in ("gemm.hip", "gemm_x3.hip"):
            h.update(open(os.path.join(ROOT, "pdfnet_amd", "csrc", f), "rb").read())
        if d.get("gemm_hip_sha256") != h.hexdigest():
            return None, None, "%s is older than csrc/gemm.hip / gemm_x3.hip (withheld)" % os.path.basename(p)
        per_symbol = {k: round(v["bytes_per_launch"]) for k, v in d.get("symbols", {}).items()}
        return (per_symbol, round(d["gemm_family"]["bytes_per_launch"]),
                "%s (rocprofv3 --pmc, %s)" % (os.path.basename(p), d.get("tag", "")))
    except Exception as e:                                     # noqa: BLE001
        return None, None, "unreadable profile: %s" % e


def pmc_traffic_bf16(batch=64):
    """{bf16 kernel symbol: HBM bytes per launch} from the last committed PMC profile's `bf16_symbols` (tools/profile_step.sh: separate
    --pmc FETCH_SIZE / WRITE_SIZE passes of `bench.py --dtype bf16 --batch 64`; `bf16_symbols_B32` for the B=32 step's own passes),
    withheld when csrc/gemm_bf16.hip / gemm_dma.hip have changed since (same rule as pmc_traffic)."""
    import glob
    import hashlib
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
    if not files:
        return None, "no PMC profile committed"
    p = files[-1]
    try:
        d = json.load(open(p))
        key = "bf16_symbols" if batch == 64 or ("bf16_symbols_B%d" % batch) not in d else "bf16_symbols_B%d" % batch
        run_b = 64 if key == "bf16_symbols" else batch
        if key not in d:
            return None, "%s has no bf16 pass" % os.path.basename(p)
        h = hashlib.sha256()
        for f in ("gemm_bf16.hip", "gemm_dma.hip"):
            h.update(open(os.path.join(ROOT, "pdfnet_amd", "csrc", f), "rb").read())
        if d.get("bf16_src_sha256") != h.hexdigest():
            return None, "%s is older than csrc/gemm_bf16.hip / gemm_dma.hip (withheld)" % os.path.basename(p)
        return {k: round(v["bytes_per_launch"]) for k, v in d[key].items()}, "%s (rocprofv3 --pmc, bf16 B=%d run)" % (os.path.basename(p), run_b)
    except Exception as e:                                     # noqa: BLE001
        return None, "unreadable profile: %s" % e


def pmc_traffic_hbm():
    """HBM bytes per step of the PointNet++ data-movement kernels from the last committed PMC profile (same staleness rule
    as pmc_traffic, keyed on csrc/pointops.hip + csrc/norm.hip)."""
    import glob
    import hashlib
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
    if not files:
        return None, "no PMC profile committed"
    p = files[-1]
    try:
        d = json.load(open(p))
        h = hashlib.sha256()
        for f in ("pointops.hip", "norm.hip"):
            h.update(open(os.path.join(ROOT, "pdfnet_amd", "csrc", f), "rb").read())
        if d.get("hbm_src_sha256") != h.hexdigest():
            return None, "%s is older than csrc/pointops.hip / norm.hip (withheld)" % os.path.basename(p)
        return round(d["hbm_family"]["pointnet_bytes_per_step"]), "%s (rocprofv3 --pmc, %s)" % (os.path.basename(p), d.get("tag", ""))
    except Exception as e:                                     # noqa: BLE001
        return None, "unreadable profile: %s" % e


def oracle_with_loss(R, state_dict=None):
    """CPU oracle model + CPU oracle CtdetLoss (oracle/pdfnet_cpu.py, oracle/loss_cpu.py: both pinned to the reference)."""
    import numpy as np
    from oracle import loss_cpu as LC
    from oracle import pdfnet_cpu as O
    from pdfnet_amd.synthetic import synthetic_loss_constants
    opt = make_opt(R)
    o = O.load_model_cpu(opt)
    if state_dict is not None:
        o.load_state_dict(state_dict)
    z = np.load(os.path.join(ROOT, "pdfnet_amd", "data", "gcn_core.npz"))
    conv = {h: LC.Converter(z['graph_perm_' + h], z['graph_perm_reverse_' + h]) for h in ('left', 'right')}
    consts = synthetic_loss_constants()

    def run(batch, mode, epoch=0):
        ind = batch['ind'] if mode == 'train' else None
        result, params, hand, other = o(batch['input'], batch['choose'], batch['cloud'], batch['depth'], ind, batch['K_new'], batch['valid'])
        other['converter_left'], other['converter_right'] = conv['left'], conv['right']
        return LC.ctdet_loss(opt, consts, result, params, hand, other, batch, mode, epoch)
    return o, run, consts


def cpu_baseline(R, threads, B=4, steps=3):
    """The CPU oracle on a bounded sample of the SAME workload: B=4, the oracle model + the oracle CtdetLoss (same loss as
    the GPU step), forward + loss + backward + torch Adam; one warm-up step, then the median of `steps` timed steps."""
    from pdfnet_amd.synthetic import synthetic_train_batch
    torch.set_num_threads(threads)
    o, run, consts = oracle_with_loss(R)
    opt = torch.optim.Adam(o.parameters(), lr=1e-4)
    b = synthetic_train_batch(B, R, seed=1, consts=consts)
    o.train()
    times = []
    t_all = time.time()
    for i in range(steps + 1):
        t0 = time.time()
        opt.zero_grad()
        loss, _ = run(b, 'train', 0)
        loss.mean().backward()
        opt.step()
        times.append(time.time() - t0)
        if i >= 1 and time.time() - t_all + times[-1] > 45.0:       # bounded: ~10-30 s of CPU work on a fast host; a slow host (33 s per step seen) stops after one timed step
            break
    steps = len(times) - 1
    med = sorted(times[1:])[len(times[1:]) // 2]
    return {"value": round(B / med, 4), "unit": "images/s", "cores": threads, "kind": "port",
            "sample": "CPU oracle (oracle/pdfnet_cpu.py + oracle/loss_cpu.py, PyTorch fp32): B=%d %dx%d RGB-D, full train step "
                      "(fwd + CtdetLoss + bwd + Adam), median of %d steps after 1 warm-up (%.1f s each)" % (B, R, R, steps, med)}


def mpjpe_report(trainer, consts, R, B_eval, dev):
    """MPJPE on synthetic ground truth (metric: "...; MPJPE parity vs ref"): the HIP path's test-mode pass over a fresh
    synthetic batch (lib/trains/base_trainer.py:207-429 counterpart, all ranks), and -- rank 0, bounded to B=2 -- the same
    weights and samples through the CPU oracle (model + loss pinned to the reference)."""
    from pdfnet_amd.synthetic import synthetic_train_batch
    rank = dist.get_rank() if dist.is_initialized() else 0
    batch = synthetic_train_batch(B_eval, R, seed=1001 + rank, consts=consts)
    ev = trainer.evaluation([batch], dev)
    out = {"mpjpe_mm": round(ev['mpjpe_mm'], 3), "mpvpe_mm": round(ev['mpvpe_mm'], 3), "mpjpe_root_relative_mm": round(ev['mpjpe_off_mm'], 3),
           "lms_px": round(ev['lms_px'], 3), "samples": ev['samples']}
    return out, batch


def mpjpe_parity_hip_half(trainer, dev, batch):
    """The HIP half of mpjpe_parity: the HIP path's metrics on the first two samples of `batch` and a CPU copy of the (trained) weights --
    taken while the fp32 trainer is alive; the CPU oracle half runs after the bf16 legs (its 128 host threads and GBs of CPU tensors in
    front of them cost the host-bound B=32 leg ~4 ms per step: 879 vs 950 img/s, r05)."""
    from pdfnet_amd.trains.base_trainer import evaluation_sums, finish_evaluation
    sub = {k: v[:2] for k, v in batch.items()}
    mwl = trainer.model_with_loss
    mwl.eval()
    with torch.no_grad():
        tup = mwl({k: v.to(dev) for k, v in sub.items()}, 'test', None)
        hip = finish_evaluation(evaluation_sums(tup, {k: v.to(dev) for k, v in sub.items()}).cpu())
    sd = {k: v.detach().cpu().contiguous() for k, v in trainer.model.state_dict().items()}
    mwl.train()
    return hip, sd, sub


def mpjpe_parity(half, R):
    """HIP path vs CPU oracle path on two samples, same (trained) weights: |delta MPJPE| in mm.  half: mpjpe_parity_hip_half(...)."""
    from oracle import loss_cpu as LC
    hip, sd, sub = half
    o, run, _ = oracle_with_loss(R, sd)
    o.eval()
    with torch.no_grad():
        ref = LC.evaluation_metrics(run(sub, 'test'), (sub['lms_left_gt'], sub['lms_right_gt']))
    ref_mpjpe = (ref['abs_left_joints'] + ref['abs_right_joints']) / 2
    return {"hip_mm": round(hip['mpjpe_mm'], 4), "cpu_oracle_mm": round(ref_mpjpe, 4), "abs_diff_mm": round(abs(hip['mpjpe_mm'] - ref_mpjpe), 5),
            "samples": 2, "note": "same weights (after the timed steps) and samples through the HIP path and the CPU oracle path"}


def rgb_encoder_bench(args, dev, rank, world):
    """BASELINE config 2: B=8 256x256 RGB-only ResNet encoder forward + backward (intaghand_encoder.py:711-744: e_conv1,
    ResNet-50, pyramid laterals + L2Norm, feat + BN + ReLU) with the weight gradients accumulated into the flat buffer --
    the MFMA conv kernels on their own.  No loss module, no optimizer: a step is forward + backward of sum(x0^2)/N + ..."""
    from pdfnet_amd import functional as F
    from pdfnet_amd.networks.intaghand_model import load_model_intag
    from pdfnet_amd.trains.base_trainer import FlatAdam
    R, B = args.res, args.batch
    torch.manual_seed(0)
    enc = load_model_intag(make_opt(R)).encoder.to(dev)
    names = ('resnet.', 'p2', 'p3', 'p4', 'p5', 'feat', 'e_conv1')
    params = [p for n, p in enc.named_parameters() if n.startswith(names) and not n.startswith('resnet.fc')]
    flat = FlatAdam(params, lr=1e-4)
    img = torch.randn(B, 3, R, R, device=dev)
    enc.train()

    def step():
        flat.zero_grad()
        x0, emb0, x1 = enc.rgb_encoder(img)
        (x0.pow(2).mean() + emb0.pow(2).mean() + x1.pow(2).mean()).backward()
        F.join_wgrad()
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    dt = time.time() - t0
    # algorithmic FLOPs of this sub-path, SURVEY 8(d): a1 10.68 + a2 15.57 + a3 19.33 + a7 0.01 GF/img forward, x3 for the step
    gf_img = 3 * (10.68 + 15.57 + 19.33 + 0.01) * (R / 256.0) ** 2
    out = {"metric": "RGB-only encoder fwd+bwd images/sec @%dx%d B=%d" % (R, R, B), "value": round(B * args.steps / dt, 2), "unit": "images/s",
           "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": "configs[1]: B=%d RGB-only ResNet-50 encoder + pyramid + feat forward/backward (weight gradients included), "
                                  "fp32, %dx%d" % (B, R, R), "global_batch": B, "parallelism": "dp1"},
           "roofline": {"bound": "mfma", "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s", "kernel": "whole sub-path (step level)",
                        "achieved": round(gf_img * B * args.steps / dt / 1e3, 2), "frac": round(gf_img * B * args.steps / dt / 1e3 / PEAK_FP32_MFMA_TFLOPS, 4),
                        "algorithmic_gflop_per_img_step": round(gf_img, 2), "traffic": None}}
    print(json.dumps(out), flush=True)


def bf16_leg(opt, R, B, dev, consts, steps, warmup):
    """BASELINE configs[3] / configs[4] as their per-GPU step (B = 32 / 64 per rank): bf16-input MFMA GEMMs
    with fp32 accumulation, bf16 shadows, fp32 master weights / statistics / loss, the trainer configured for bf16 gradient
    transport.  A fresh model and trainer, timed like the headline (warm-up, K steps between synchronisations), then one
    instrumented step for the dominant kernel symbol."""
    from pdfnet_amd import functional as F
    from pdfnet_amd.networks.intaghand_model import load_model_intag
    from pdfnet_amd.synthetic import synthetic_train_batch, to_device
    from pdfnet_amd.trains.base_trainer import Trainer
    from pdfnet_amd.trains.simplified import CtdetLoss
    F.set_gemm_precision('bf16')
    try:
        import gc
        gc.collect()
        torch.cuda.empty_cache()                             # the previous run's cached blocks have other sizes: start from a clean pool
        torch.manual_seed(0)
        F.manual_seed(4321)
        model = load_model_intag(opt).to(dev)
        # use_graph='auto': the trainer times a few eager and a few hipGraph steps during the warm-up and keeps the faster mode (the B=32
        # step is host-bound on slow hosts, where the graph wins; B=64 stays eager)
        tr = Trainer(opt, model, CtdetLoss(opt, consts).to(dev), lr=1e-4, grad_comm_dtype=torch.bfloat16, use_graph='auto')
        warmup = max(warmup, 2 * (Trainer.AUTO_SKIP + Trainer.AUTO_STEPS + 1) + 1)
        batch = to_device(synthetic_train_batch(B, R, seed=1, consts=consts), dev)
        for _ in range(warmup):
            tr.train_step(batch)
        torch.cuda.synchronize()
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
        # the B=32 step is within 10 % of the host's issue time: keep the cyclic collector (everything the fp32 run, the CPU oracle
        # and the MPJPE passes left behind is tracked) out of the timed region, as a training loop on a busy host would
        gc.collect()
        gc.freeze()
        gc.disable()
        try:
            # the collection above destroys what the launch-mode trial left behind (the losing mode's hipGraph and its private memory
            # pool: tens of GB of hipFree), and the first steps after it re-grow the allocator's cache -- r04: a 1.6 s stall INSIDE the
            # timed loop of the B=64 leg (226 ms/step at a median of 62.7), r03: 75 vs 60 ms.  Settle before the clock starts.
            for _ in range(8):
                tr.train_step(batch)
                torch.cuda.synchronize()
            # timed ONCE (round 6: the window is never replaced -- a stalled step stays in `images_per_s`; `longest_step_ms` and the median say so)
            t0 = time.time()
            marks[0].record()
            for i in range(steps):
                last = tr.train_step(batch)
                marks[i + 1].record()                      # (events on the launch stream: per-step times without a host sync)
            torch.cuda.synchronize()
            dt = time.time() - t0
        finally:
            gc.enable()
            gc.unfreeze()
        per_step = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(steps))
        median_ms = per_step[steps // 2]
        loss_val = float(last)
        assert loss_val == loss_val, "bf16 leg: loss is NaN"
        launch = "hipGraph" if tr.use_graph else "eager"
        choice = getattr(tr, 'auto_choice', None)
        tr.collectives = False
        tr.use_graph = False                                   # (the instrumented step launches kernel by kernel)
        F.USE_SIDE_STREAMS = False
        with KernelTimer() as kt:
            tr.train_step(batch)
        sym = kt.by_symbol()
        F.USE_SIDE_STREAMS = True
        tr16, tr16_src = pmc_traffic_bf16(B)
        head, _ = symbol_roofline(sym, PEAK_BF16_MFMA_TFLOPS, tr16, tr16_src)
        fl, sec = sum(v[1] for v in sym.values()), sum(v[3] for v in sym.values())
        out = {"images_per_s": round(B * steps / dt, 2), "ms_per_step": round(dt / steps * 1e3, 3),
               "median_step_ms": round(median_ms, 3), "images_per_s_at_median_step": round(B / median_ms * 1e3, 2),
               "longest_step_ms": round(per_step[-1], 3), "batch": B, "steps": steps, "warmup": warmup, "launch": launch,
               "auto_choice_ms": {k: round(v, 3) for k, v in choice.items()} if choice else None,
               "final_loss": round(loss_val, 4),
               "dominant_kernel": {k: head[k] for k in ("kernel", "achieved", "frac", "launches_per_step", "ms_per_step", "algorithmic_bytes_per_launch", "traffic", "traffic_source")},
               "all_gemm_kernels_tflops": round(fl / max(sec, 1e-9) / 1e12, 1), "gemm_ms_per_step_exclusive": round(sec * 1e3, 2),
               "step_level_frac_of_bf16_mfma_peak": round(fl / B * (B * steps / dt) / 1e12 / PEAK_BF16_MFMA_TFLOPS, 4)}
        del tr, model, batch
        torch.cuda.empty_cache()
        return out
    finally:
        F.USE_SIDE_STREAMS = True
        F.set_gemm_precision('fp32')


def check_grads(trainer, batch, world, dev):
    """--check-grads (N > 1): the gradient the trainer's overlapped all-reduce leaves in the flat buffer must equal the
    sum over ranks of the rank-local gradients, gathered with a plain all_gather.  Two extra steps after the timed region
    at frozen weights (lr = 0) with dropout off, so both see the same function."""
    tr = trainer
    tr.optimizer.lr = 0.0
    for mod in tr.model.modules():
        if isinstance(getattr(mod, 'p', None), float):
            mod.p = 0.0
    tr.collectives = False
    tr.train_step(batch)                                 # rank-local gradient
    torch.cuda.synchronize()
    local = tr.optimizer.flat_g[:tr.n_live].clone()
    parts = [torch.empty_like(local) for _ in range(world)]
    dist.all_gather(parts, local)
    want = torch.stack(parts).sum(0)
    tr.collectives = True
    tr.train_step(batch)                                 # early slice reduced from inside the backward, late slice after it
    torch.cuda.synchronize()
    got = tr.optimizer.flat_g[:tr.n_live]
    err = float((got - want).abs().max() / (want.abs().max() + 1e-30))
    return {"max_abs_err_over_max_abs": err, "ok": bool(err < 1e-4), "elements": int(got.numel()), "ranks": world}


def collective_path(trainer, batch, steps=10, warmup=3):
    """What the data-parallel reduction path costs ONE rank before a byte crosses xGMI: the same train step with
    the autograd hook active and both gradient all-reduces (early slice from inside the backward on the communication stream, late
    slice after it) issued for real on a ONE-rank RCCL group, against the plain step -- interleaved plain / fp32 transport / plain /
    bf16 transport / plain, medians of per-step HIP-event times.  N = 1 only (at N > 1 the timed region IS this path)."""
    created = False
    if not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', str(free_port()))
        dist.init_process_group('nccl', rank=0, world_size=1)
        created = True

    def med(force, comm_dtype):
        trainer.force_collectives = force
        trainer.reducer.comm_dtype = comm_dtype
        for _ in range(warmup):
            trainer.train_step(batch)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
        ev[0].record()
        for i in range(steps):
            trainer.train_step(batch)
            ev[i + 1].record()
        torch.cuda.synchronize()
        return sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(steps))[steps // 2]
    keep = trainer.reducer.comm_dtype
    try:
        p0 = med(False, None)
        f32 = med(True, None)
        p1 = med(False, None)
        b16 = med(True, torch.bfloat16)
        p2 = med(False, None)
    finally:
        trainer.force_collectives = False
        trainer.reducer.comm_dtype = keep
        if created:
            dist.destroy_process_group()
    return {"plain_step_ms": [round(p0, 3), round(p1, 3), round(p2, 3)], "fp32_transport_step_ms": round(f32, 3), "bf16_transport_step_ms": round(b16, 3),
            "collective_path_ms": round(f32 - (p0 + p1) / 2, 3), "collective_path_bf16_transport_ms": round(b16 - (p1 + p2) / 2, 3),
            "steps": steps, "note": "one-rank RCCL group (force_collectives): hook + stream plumbing + two all-reduce launches (+ the bf16 cast and "
                                    "widening passes); median step time minus the mean of the neighbouring plain medians"}


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def launcher_command(n, argv, port=None):
    """The command line `python bench.py --gpus N` turns itself into: one rank per GPU under torch.distributed.run, rendezvous on
    127.0.0.1 (scripts/train.sh:21 + main.py:69-71 of the reference: `torch.distributed.launch --nproc_per_node N main.py`)."""
    return [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n), '--master-addr', '127.0.0.1',
            '--master-port', str(port or free_port()), os.path.abspath(__file__)] + list(argv)


def launch_ranks(n, argv, run=None, device_count=None):
    """`python bench.py --gpus N` without torch.distributed.run around it: start the N ranks as a CHILD process (a process that
    has initialised the GPU must never be replaced by exec, and this one has not initialised anything: torch.cuda.device_count()
    only counts), relay rank 0's JSON line and the child's return code.  -> exit code."""
    import subprocess
    have = torch.cuda.device_count() if device_count is None else device_count
    if have < n:
        print("bench.py: --gpus %d asked for, %d GPU(s) visible on this machine: not launching" % (n, have), file=sys.stderr, flush=True)
        return 2
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')          # dmabuf IPC: RCCL across processes needs it on this pool
    env.setdefault('OMP_NUM_THREADS', str(max(1, (os.cpu_count() or n) // n)))
    cmd = launcher_command(n, argv)
    print("bench.py: launching %d ranks: %s" % (n, ' '.join(cmd)), file=sys.stderr, flush=True)
    r = (run or subprocess.run)(cmd, env=env)                  # the child's stdout IS ours: rank 0's one JSON line passes through
    return r.returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--config', default='full', choices=['full', 'rgb-encoder'],
                    help="full: BASELINE configs[2] (the headline); rgb-encoder: configs[1], B=8 RGB-only ResNet encoder fwd/bwd")
    ap.add_argument('--batch', type=int, default=None, help='per-GPU batch (default 32; rgb-encoder: 8)')
    ap.add_argument('--res', type=int, default=256)
    ap.add_argument('--dtype', default='f32', choices=['f32', 'bf16'],
                    help="f32: BASELINE configs[2] (headline). bf16: configs[3]/[4] per-GPU step -- bf16-input MFMA GEMMs with fp32 accumulation, "
                         "fp32 master weights / statistics / loss, bf16 gradient all-reduce")
    ap.add_argument('--graph', action='store_true', help='replay forward+loss+backward as one hipGraph (default: eager launches with\n'
                    'side-stream overlap of the weight-gradient kernels, measured faster on MI355X)')
    ap.add_argument('--no-graph', action='store_true', help='(default) kept for compatibility')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--no-mpjpe', action='store_true')
    ap.add_argument('--broadcast-buffers', action='store_true', help="DDP's per-iteration BN-buffer broadcast (base_trainer.py:94-95)")
    ap.add_argument('--check-grads', action='store_true', help='(default for --gpus > 1) verify the reduced gradient against an all_gather of the rank-local ones')
    ap.add_argument('--no-check-grads', action='store_true')
    ap.add_argument('--no-collective-path', action='store_true', help='skip the one-rank RCCL measurement of the reduction path (config.collective_path_ms)')
    ap.add_argument('--no-native-leg', action='store_true', help='skip the all-native-fp32-MFMA steps timed beside the shipped (x3) step (arithmetic.native_fp32_mfma_step)')
    ap.add_argument('--no-bf16-legs', action='store_true', help='skip the bf16 B=32 / B=64 per-GPU legs (bf16_per_gpu) of the default fp32 run')
    ap.add_argument('--gemm-shapes', default=None, help='write the per-shape table of the instrumented step to this file')
    ap.add_argument('--hbm-shapes', default=None, help='the same for the HBM-bound entry points (integer arguments, algorithmic MB, ms, GB/s)')
    args = ap.parse_args()
    if args.batch is None:
        args.batch = 8 if args.config == 'rgb-encoder' else 32
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # started directly (`python bench.py --gpus N`): this process becomes the launcher.  Nothing here has touched the GPU.
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))

    from pdfnet_amd import functional as F
    from pdfnet_amd.networks.intaghand_model import load_model_intag
    from pdfnet_amd.synthetic import synthetic_loss_constants, synthetic_train_batch, to_device
    from pdfnet_amd.trains.base_trainer import Trainer, init_distributed
    from pdfnet_amd.trains.simplified import CtdetLoss

    # RCCL writes its NCCL_DEBUG=VERSION banner to the C stdout buffer, which is flushed at exit -- AFTER the JSON line the
    # driver reads.  Keep rank 0's stdout to that one line: drop the banner (a user's INFO / TRACE setting is left alone) ...
    if os.environ.get('NCCL_DEBUG', '').upper() == 'VERSION':
        os.environ['NCCL_DEBUG'] = 'WARN'
    rank, local, world = init_distributed()
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d (launch with --nproc-per-node %d, or plain `python bench.py --gpus %d`)"
                 % (args.gpus, world, args.gpus, args.gpus))
    dev = torch.device('cuda', local)
    if args.config == 'rgb-encoder':
        assert world == 1, "--config rgb-encoder is a one-GPU kernel benchmark"
        return rgb_encoder_bench(args, dev, rank, world)
    R, B = args.res, args.batch
    opt = make_opt(R)
    bf16 = args.dtype == 'bf16'
    if bf16:
        F.set_gemm_precision('bf16')
    torch.manual_seed(0)
    F.manual_seed(1234 + rank)
    model = load_model_intag(opt).to(dev)
    consts = synthetic_loss_constants()
    loss = CtdetLoss(opt, consts).to(dev)
    if args.graph and os.environ.get('PDF_GRAPH_WGRAD_STREAM') == '0':
        F.ASYNC_WGRAD = False              # (the weight-gradient branch inside the capture: grouped launches, Trainer.GRAPH_WGRAD_GROUP -- r05_wgrad_group.txt)
    trainer = Trainer(opt, model, loss, lr=1e-4, use_graph=args.graph, broadcast_buffers=args.broadcast_buffers,
                      grad_comm_dtype=torch.bfloat16 if bf16 else None)   # world > 1: replicas synced from rank 0
    batch = to_device(synthetic_train_batch(B, R, seed=1 + rank, consts=consts), dev)
    rccl_ranks = 1
    if world > 1:
        ones = torch.ones(1, device=dev)
        dist.all_reduce(ones)                                  # what the data path's collective sees: every rank answers
        rccl_ranks = int(ones.item())

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        trainer.train_step(batch)
    barrier()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.time()
    marks[0].record()
    for i in range(args.steps):
        last = trainer.train_step(batch)
        marks[i + 1].record()                              # events on the launch stream: per-step times without a host sync
    barrier()
    dt = time.time() - t0
    per_step = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))
    median_ms = per_step[args.steps // 2]
    if world > 1:
        t = torch.tensor([dt], device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
    loss_val = float(last)
    assert loss_val == loss_val, "loss is NaN"

    out = {
        "metric": "train-step images/sec @256x256 RGB-D B=32", "value": round(world * B * args.steps / dt, 2),
        "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 3), "median_step_ms": round(median_ms, 3),
        "images_per_s_at_median_step": round(world * B / median_ms * 1e3, 2), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": "configs[%s]: B=%d/GPU full RGB-D pyramid fusion + PointNet++ + GCN decoder fwd + CtdetLoss + bwd + Adam, "
                               "%s, %dx%d" % ("3/4" if bf16 else "2", B, "bf16 MFMA GEMMs (fp32 accumulate, fp32 master weights / statistics / loss)" if bf16 else "fp32", R, R), "global_batch": world * B, "parallelism": "dp%d" % world,
                   "launch": "hipGraph(fwd+loss+bwd) + fused Adam" if args.graph else "eager, weight-gradient kernels overlapped on a side HIP stream, fused Adam", "final_loss": round(loss_val, 4),
                   "rccl_ranks": rccl_ranks, "allreduce_mb_per_step": round(trainer.n_live * (2 if bf16 else 4) / 1e6, 1) if world > 1 else 0.0,
                   "broadcast_buffers": bool(args.broadcast_buffers),
                   # what pdfnet_amd/taped.py keeps alive for the life of the process (the trunk's activations of every recorded signature)
                   "taped_pinned_mb": round(getattr(model.encoder.__dict__.get('_trunk_seg'), 'pinned_bytes', lambda: 0)() / 1e6, 1)},
    }
    x3_mode = F._L().pdf_debug_x3_mode() if not bf16 else 0
    if x3_mode:
        # the layers that run as x3 products are named, and the all-native-fp32-MFMA step is timed beside the shipped one (same
        # trainer, same batch, x3 switched off at run time; a third of the steps, not part of `value`)
        out["arithmetic"] = {
            "x3_mode": x3_mode,
            "x3_layers": "fp32 via 3 x bf16 split (six bf16 MFMAs per fp32 product), fp32 accumulate: " + ", ".join(
                (["the Winograd F(4x4)-domain products of the stride-1 3x3 layers with >= 256 channels on >= 2,048 tiles -- at B = 32: `feat` 1024 -> 256, p2 and the "
                  "hm / wh / params head convolutions 256 -> 256, all on the 64x64 map (forward, backward-data, weight gradient)"] if x3_mode & 1 else []) +
                (["p3 / p4 / p5 transposed convolutions (forward, backward-data, weight gradient)"] if x3_mode & 2 else []) +
                (["the linear products of the fused mesh decoder (GCN blocks, attention projections and MLPs, forward and data gradients; operands split in registers)"] if x3_mode & 4 else [])),
            "everything_else": "native fp32 MFMA (v_mfma_f32_32x32x2_f32)",
            "error_vs_float64": "x3 products 0.31-0.37x the native kernel's rms error, max error 0.35-0.45x (profiles/r06_x3_bench.txt; enforced by tests/test_x3_gpu.py)"}
        if world == 1 and not args.graph and not args.no_native_leg:
            try:
                F.set_x3(0)
                kn = max(5, args.steps // 3)
                for _ in range(3):
                    trainer.train_step(batch)
                barrier()
                tn0 = time.time()
                for _ in range(kn):
                    trainer.train_step(batch)
                barrier()
                dtn = time.time() - tn0
                out["arithmetic"]["native_fp32_mfma_step"] = {"images_per_s": round(B * kn / dtn, 2), "ms_per_step": round(dtn / kn * 1e3, 3), "steps": kn}
            except Exception as e:                             # noqa: BLE001 -- the headline line must still be printed
                out["arithmetic"]["native_fp32_mfma_step"] = {"error": "%s: %s" % (type(e).__name__, e)}
            finally:
                F.set_x3(None)
                for _ in range(2):
                    trainer.train_step(batch)                  # (the instrumented step below runs the shipped configuration, allocator warm)
                barrier()
    if world == 1 and not bf16 and not args.graph and not args.no_collective_path:
        try:
            cp = collective_path(trainer, batch)
            out["config"]["collective_path_ms"] = cp["collective_path_ms"]
            out["config"]["collective_path"] = cp
        except Exception as e:                                 # noqa: BLE001 -- the headline line must still be printed
            out["config"]["collective_path"] = {"error": "%s: %s" % (type(e).__name__, e)}
    mp_batch = None
    if not args.no_mpjpe:
        out["mpjpe"], mp_batch = mpjpe_report(trainer, consts, R, min(B, 8), dev)      # every rank takes part (all-reduced sums)
    if world > 1 and not args.no_check_grads:
        out["check_grads"] = check_grads(trainer, batch, world, dev)
    if rank == 0 and not args.no_roofline:
        # instrumented eager step: events around every implicit-GEMM entry point on the launch stream
        trainer.use_graph = False
        trainer.collectives = False        # rank-local step: the other ranks are already past their last collective
        F.USE_SIDE_STREAMS = False         # exclusive per-launch durations (no overlapped branches)
        with GemmProfiler() as prof, HbmProfiler() as hprof, KernelTimer() as ktimer:
            trainer.train_step(batch)
        sym = ktimer.by_symbol()
        per = prof.summary()
        hper = hprof.summary()
        if args.gemm_shapes:
            with open(args.gemm_shapes, 'w') as f:
                for (n, ints), (c, fl, sec) in prof.by_shape():
                    f.write("%-26s %-60s calls %3d  %8.3f ms  %6.1f TF\n" % (n, ' '.join(map(str, ints)), c, sec * 1e3, fl / max(sec, 1e-9) / 1e12))
        if args.hbm_shapes:
            with open(args.hbm_shapes, 'w') as f:
                for (n, ints), (c, nb, sec) in hprof.by_shape():
                    f.write("%-26s %-48s calls %3d  %9.1f MB  %8.3f ms  %7.1f GB/s\n" % (n, ' '.join(map(str, ints)), c, nb / 1e6, sec * 1e3, nb / max(sec, 1e-9) / 1e9))
        calls = sum(v[0] for v in per.values())
        flops = sum(v[1] for v in per.values())
        secs = sum(v[2] for v in per.values())
        peak = PEAK_BF16_MFMA_TFLOPS if bf16 else PEAK_FP32_MFMA_TFLOPS
        if bf16:
            traffic, traffic_src = pmc_traffic_bf16(B)
            traffic_all = None
        else:
            traffic, traffic_all, traffic_src = pmc_traffic()
        # the rocprofv3 symbol with the most time per step, timed per kernel launch by the library itself (events on the launch
        # stream): name, launches and ms are one row of profiles/*_kernel_stats_exclusive.csv
        head, table = symbol_roofline(sym, peak, traffic, traffic_src)
        exec_flops = sum(v[1] for v in sym.values())
        # the fused mesh decoder's own products (forward, data gradients: csrc/meshdec.hip) are not KernelTimer symbols; its weight-gradient
        # launches are.  Add the former: (fwd) lin + att, (bwd) lin + 2.5 att per level.
        mesh_fl = 0.0
        for n, fl, _e0, _e1, ints, t, _l in prof.records:
            if t == -2:
                lin, att = GemmProfiler.mesh_level_flops(*ints)
                mesh_fl += (lin + att) if n.endswith('fwd') else (lin + 2.5 * att)
        exec_flops += mesh_fl
        out["roofline"] = {
            "bound": "mfma", "peak": peak, "unit": "TFLOP/s", **head,
            "per_symbol": table,
            "all_gemm_kernels": {
                "achieved": round(flops / secs / 1e12, 2), "frac": round(flops / secs / 1e12 / peak, 4),
                "launches_per_step": calls, "algorithmic_gflop_per_step": round(flops / 1e9, 1), "gemm_ms_per_step": round(secs * 1e3, 2),
                "traffic": traffic_all,
                "per_entry_point": {k: {"calls": v[0], "gflop": round(v[1] / 1e9, 1), "ms": round(v[2] * 1e3, 2),
                                        "tflops": round(v[1] / max(v[2], 1e-9) / 1e12, 1)} for k, v in sorted(per.items())}},
            "formulation": "all_gemm_kernels and step_level.algorithmic_* count ALGORITHMIC contraction FLOPs from the entry points' arguments (2 M N K of the direct "
                           "sum, exact sparse centre features, SURVEY 8a6): %.0f GFLOP/img/step; reference formulation (dense centre convs) = 358.  " % (flops / 1e9 / B) +
                           "The stride-1 3x3 layers with >= 128 channels EXECUTE 4x / 2.25x fewer multiplications (Winograd F(4x4,3x3) / F(2x2,3x3), "
                           "csrc/winograd.hip, fp32): per_symbol, the dominant-kernel figures and step_level.executed_frac count what each kernel executes",
            "step_level": {"gflop_per_img_step_reference_formulation": ALGO_GFLOP_PER_IMG_STEP_DENSE,
                           "tflops_reference_formulation": round(ALGO_GFLOP_PER_IMG_STEP_DENSE * out["value"] / world / 1e3, 2),
                           "gflop_per_img_step_algorithmic": round(flops / 1e9 / B, 1),
                           "tflops_algorithmic": round(flops / 1e9 / B * out["value"] / world / 1e3, 2),
                           # direct-sum FLOPs of the path over the step time: an EQUIVALENT rate (Winograd layers execute 4x / 2.25x fewer)
                           "algorithmic_equivalent_frac": round(flops / 1e9 / B * out["value"] / world / 1e3 / peak, 4),
                           # what the matrix pipe really executes per step (sum of per_symbol) over the step time: MFMA utilisation
                           "gflop_per_step_executed": round(exec_flops / 1e9, 1),
                           "tflops_executed": round(exec_flops / (dt / args.steps) / 1e12, 2),
                           "executed_frac": round(exec_flops / (dt / args.steps) / 1e12 / peak, 4)},
        }
        # kNN + ball query moves 12 MB per step and is bound by its selection (every centroid ranks every point of its cloud and keeps 64):
        # it is reported against that work, not against HBM, and stays out of the HBM aggregates
        knn = hper.pop('pdf_knn_ball_group', None)
        hb, hs = sum(v[1] for v in hper.values()), sum(v[2] for v in hper.values())
        pn = ('pdf_gather_sub_fwd', 'pdf_gather_sub_bwd', 'pdf_gather_sub_bwd_sorted', 'pdf_bn_relu_maxk_fwd', 'pdf_bn_relu_maxk_bwd', 'pdf_gather_rows', 'pdf_scatter_rows_add')
        pb, ps = sum(hper[k][1] for k in pn if k in hper), sum(hper[k][2] for k in pn if k in hper)
        htraffic, htraffic_src = pmc_traffic_hbm()
        out["roofline_hbm"] = {
            "bound": "hbm", "peak": HbmProfiler.PEAK_GBS, "unit": "GB/s",
            "kernel": "PointNet++ data movement: per-point conv gather (gather_sub), set-abstraction tail "
                      "(BatchNorm + ReLU + max over K in one pass), pyramid row gathers -- csrc/pointops.hip, csrc/norm.hip",
            "achieved": round(pb / max(ps, 1e-9) / 1e9, 1), "frac": round(pb / max(ps, 1e-9) / 1e9 / HbmProfiler.PEAK_GBS, 4),
            "launches_per_step": sum(hper[k][0] for k in pn if k in hper), "ms_per_step": round(ps * 1e3, 3),
            "algorithmic_MB_per_step": round(pb / 1e6, 1), "traffic": htraffic, "traffic_source": htraffic_src,
            "all_hbm_bound_entry_points": {"achieved": round(hb / max(hs, 1e-9) / 1e9, 1), "frac": round(hb / max(hs, 1e-9) / 1e9 / HbmProfiler.PEAK_GBS, 4),
                                           "ms_per_step": round(hs * 1e3, 2), "algorithmic_GB_per_step": round(hb / 1e9, 2)},
            "per_entry_point": {k: {"calls": v[0], "MB": round(v[1] / 1e6, 1), "ms": round(v[2] * 1e3, 3), "GBs": round(v[1] / max(v[2], 1e-9) / 1e9, 1)}
                                for k, v in sorted(hper.items())},
            "knn_ball_group": None if knn is None else {
                "bound": "selection (VALU): each centroid ranks the N points of its cloud and keeps the K = 64 nearest inside the ball",
                "calls": knn[0], "ms": round(knn[2] * 1e3, 3), "candidate_pairs_per_step": int(hprof.knn_pairs),
                "giga_pairs_per_s": round(hprof.knn_pairs / max(knn[2], 1e-9) / 1e9, 1), "MB": round(knn[1] / 1e6, 1)},
            "fps": fps_hbm(dev),
            "fps_single_wave": fps_hbm(dev, Bc=64, N=1024, S=512),   # the reference's SAMPLE_NUM / sample_num_level1 (opts.py:226-228): one wave per cloud
        }
        F.USE_SIDE_STREAMS = True
    want_cpu = rank == 0 and world == 1 and not args.no_cpu_baseline
    parity_half = mpjpe_parity_hip_half(trainer, dev, mp_batch) if (want_cpu and mp_batch is not None) else None
    if rank == 0 and world == 1 and not bf16 and not args.no_bf16_legs and args.batch == 32:
        # BASELINE configs[3] / [4] per GPU, driver-timed in the same run: short legs after the fp32 headline.
        # The fp32 model, trainer and batches go first (measured: with them alive the legs ran 8-13 ms per step slower than the
        # same step in a fresh process -- twice the tracked Python objects for the collector, twice the live allocations)
        import gc
        del trainer, model, loss, batch, mp_batch, last
        gc.collect()
        torch.cuda.empty_cache()
        legs = {"note": "per-rank step of configs[3] (B=32/GPU) and configs[4] (B=64/GPU): bf16 MFMA GEMMs + bf16 shadows, fp32 "
                        "accumulate / master weights / statistics / loss; one GPU, no collective; launch mode chosen by Trainer(use_graph='auto')"}
        for name, b, k in (("B32", 32, 30), ("B64", 64, 20)):
            try:
                legs[name] = bf16_leg(opt, R, b, dev, consts, k, 6)
            except Exception as e:                             # noqa: BLE001 -- the fp32 headline line must still be printed
                legs[name] = {"error": "%s: %s" % (type(e).__name__, e)}
        out["bf16_per_gpu"] = legs
    if want_cpu:                                               # the CPU oracle last: nothing host-bound is timed after it
        threads = max(1, (os.cpu_count() or 2) // 2)
        out["cpu_baseline"] = cpu_baseline(R, threads)
        if parity_half is not None:
            out["mpjpe"]["parity_vs_cpu_oracle"] = mpjpe_parity(parity_half, R)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    _flush_c_stdout()                                  # ... and whatever native code buffered goes out before the result line
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
