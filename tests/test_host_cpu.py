"""CPU (-m "not gpu"): host logic of the product package -- the C-ABI library loads and exports every
symbol include/pdfnet_hip.h declares, the module tree has the reference's state_dict keys, and the product
refuses to compute without a GPU (no CPU fallback)."""
import ctypes
import os

import pytest
import torch

from tests.util import make_opt, ROOT


def test_library_exports_every_declared_symbol():
    from pdfnet_amd import hip
    protos = hip.parse_header(os.path.join(ROOT, "include", "pdfnet_hip.h"))
    assert len(protos) >= 40
    cdll = ctypes.CDLL(hip.LIB_PATH)
    for name in protos:
        assert hasattr(cdll, name), name
    # and nothing is exported that the header does not declare
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", hip.LIB_PATH], capture_output=True, text=True).stdout
    exported = {l.split()[-1] for l in out.splitlines() if " T " in l and l.split()[-1].startswith("pdf_")}
    assert exported == set(protos), exported ^ set(protos)


def test_state_dict_keys_match_reference(golden_dir):
    from pdfnet_amd.networks.intaghand_model import load_model_intag
    m = load_model_intag(make_opt(256))
    lines = [l.split() for l in open(os.path.join(golden_dir, "state_dict_manifest.txt")) if not l.startswith("#")]
    sd = m.state_dict()
    assert [l[0] for l in lines] == list(sd.keys())
    for name, shape, dtype in lines:
        want = tuple(int(s) for s in shape.split("x")) if shape != "scalar" else ()
        assert tuple(sd[name].shape) == want, name
        assert str(sd[name].dtype) == "torch." + dtype, name
    assert sum(v.numel() for v in sd.values()) == 100435929
    for attr in ("encoder", "mid_model", "decoder"):
        assert hasattr(m, attr)
    # graph tables are non-persistent, like the reference's graph_L (model_attn/gcn.py:79-86)
    assert not any("ell_" in k or "graph_L" in k for k in sd)


def test_product_has_no_cpu_fallback():
    from pdfnet_amd import functional as F
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        F.linear(torch.zeros(4, 16), torch.zeros(8, 16))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        F.conv2d(torch.zeros(1, 16, 4, 4), torch.zeros(8, 16, 3, 3))


def test_product_does_not_import_oracle():
    import re
    bad = []
    for root, _, files in os.walk(os.path.join(ROOT, "pdfnet_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(root, f)).read()
                if re.search(r"^\s*(from|import)\s+oracle\b", src, re.M) or "/root/reference" in src:
                    bad.append(f)
    assert not bad, bad


def test_checkpoint_roundtrip_with_reference_conventions(tmp_path):
    """save_model / load_model counterparts (reference lib/utils/utils.py:37-119): `module.` prefix stripped, shape
    mismatches keep the model's tensor, saved tensors are plain contiguous OIHW so the reference can read them."""
    from pdfnet_amd.networks.intaghand_model import load_model_intag
    from pdfnet_amd.utils import load_model, save_model
    torch.manual_seed(1)
    a = load_model_intag(make_opt(256))
    p = str(tmp_path / "model_5.pth")
    save_model(p, 5, a)
    ck = torch.load(p)
    assert ck['epoch'] == 5 and len(ck['state_dict']) == 1287
    assert all(v.is_contiguous() for v in ck['state_dict'].values())
    # a DDP-style checkpoint with one wrong-shaped tensor
    ck2 = {'state_dict': {('module.' + k): v for k, v in ck['state_dict'].items()}}
    ck2['state_dict']['module.encoder.hm.2.bias'] = torch.zeros(7)
    p2 = str(tmp_path / "ddp.pth")
    torch.save(ck2, p2)
    torch.manual_seed(2)
    b = load_model_intag(make_opt(256))
    keep = b.state_dict()['encoder.hm.2.bias'].clone()
    load_model(b, p2, verbose=False)
    sa, sb = a.state_dict(), b.state_dict()
    for k in sa:
        if k == 'encoder.hm.2.bias':
            assert torch.equal(sb[k], keep)
        else:
            assert torch.equal(sa[k], sb[k]), k


def test_never_used_parameter_rule_matches_the_reference_golden(golden_dir):
    """Trainer lays the parameters no autograd path reaches behind the reduced / optimised range.  The static
    name rule must select exactly the tensors that got no gradient when the reference model was run with a loss touching
    every output (tests/golden/params_without_grad.txt, written by oracle/make_goldens.py from the reference itself)."""
    from pdfnet_amd.networks.intaghand_model import load_model_intag
    from pdfnet_amd.trains.base_trainer import split_parameters
    m = load_model_intag(make_opt(256))
    named = list(m.named_parameters())
    early, late, dead = split_parameters(named)
    want = {l.strip() for l in open(os.path.join(golden_dir, "params_without_grad.txt")) if l.strip()}
    assert {n for n, _ in dead} == want
    assert len(early) + len(late) + len(dead) == len(named)
    assert sum(p.numel() for _, p in dead) == 12_387_365          # 49.5 MB of the 401 MB stay out of the all-reduce


def test_bf16_shadow_bookkeeping_ignores_stale_or_relaid_out_copies():
    """functional.attach_shadow / shadow_of (host logic of the bf16 shadows): a copy is handed to the library only while the
    tensor it mirrors is unmodified and has the same shape and strides."""
    import torch
    from pdfnet_amd import functional as F
    t = torch.randn(2, 8, 4, 4).contiguous(memory_format=torch.channels_last)
    assert F.shadow_of(t) is None and F.shadow_of(None) is None
    s = t.to(torch.bfloat16)
    F.attach_shadow(t, s)
    assert F.shadow_of(t) is None                          # fp32 mode: shadows are never handed out
    old = F._GEMM_BF16
    F._GEMM_BF16 = True                                    # (what set_gemm_precision('bf16') caches)
    try:
        assert F.shadow_of(t) is s
        t.add_(1.0)                                        # modified in place: the copy is stale
        assert F.shadow_of(t) is None
        F.attach_shadow(t, s.contiguous())                 # NCHW-contiguous copy of a channels_last tensor: other strides
        assert F.shadow_of(t) is None
        u = t.clone()
        assert F.shadow_of(u) is None                      # attributes do not travel to new tensors
        # as_matrix: the [Cout, KH*KW*Cin] view of a channels_last weight carries the gradient alias only when it IS a view
        w = torch.nn.Parameter(torch.randn(8, 4, 3, 3).contiguous(memory_format=torch.channels_last))
        w.grad = torch.zeros(8, 4, 3, 3).contiguous(memory_format=torch.channels_last)
        w._pdf_main_grad = True
        w2 = F.as_matrix(w)
        assert w2.data_ptr() == w.data_ptr() and w2._pdf_grad_alias.data_ptr() == w.grad.data_ptr()
        w.grad = torch.zeros(8, 4, 3, 3)                   # an NCHW-contiguous gradient: reshape would copy -> no alias, no tag
        assert not getattr(F.as_matrix(w), '_pdf_main_grad', False)
        wn = torch.nn.Parameter(torch.randn(8, 4, 3, 3))   # an NCHW-contiguous weight: as_matrix is a copy, autograd carries dw
        wn._pdf_main_grad = True
        wn.grad = torch.zeros(8, 4, 3, 3)
        assert not getattr(F.as_matrix(wn), '_pdf_main_grad', False)
    finally:
        F._GEMM_BF16 = old


def test_distributed_shard_is_the_reference_samplers_index_stream():
    """main.py:79,108: DistributedSampler(train_dataset) + set_epoch -- `distributed_shard` must draw the same indices, for
    ragged sizes, both drop_last modes, with and without shuffling; `ShardedLoader` then cuts batches with the loader's
    drop_last=True (main.py:88)."""
    from torch.utils.data.distributed import DistributedSampler
    from pdfnet_amd.trains.sampler import ShardedLoader, distributed_shard
    for n in (1, 7, 64, 101):
        ds = list(range(n))
        for world in (1, 2, 3, 8):
            for drop in (False, True):
                for shuffle in (True, False):
                    for epoch in (0, 3):
                        for rank in range(world):
                            ref = DistributedSampler(ds, num_replicas=world, rank=rank, shuffle=shuffle, seed=0, drop_last=drop)
                            ref.set_epoch(epoch)
                            assert list(ref) == distributed_shard(n, rank, world, epoch, 0, shuffle, drop), (n, world, drop, shuffle, epoch, rank)
    data = {'x': torch.arange(101.0).reshape(101, 1), 'meta': [str(i) for i in range(101)]}
    seen = []
    for rank in range(2):
        ld = ShardedLoader(data, batch_size=8, rank=rank, world=2, seed=0)
        ld.set_epoch(5)
        assert len(ld) == 51 // 8                            # 102 padded picks -> 51 per rank -> 6 full batches, the 7th (3 picks) dropped
        for b in ld:
            assert b['x'].shape == (8, 1) and len(b['meta']) == 8 and [int(v) for v in b['x'][:, 0]] == [int(m) for m in b['meta']]
            seen.append((rank, [int(v) for v in b['x'][:, 0]]))
    a = sum((v for r, v in seen if r == 0), [])
    b = sum((v for r, v in seen if r == 1), [])
    assert len(a) == len(b) == 48 and not set(a) & set(b)


def test_evaluation_writers_use_the_reference_formats(tmp_path):
    """H2O-val.txt block (base_trainer.py:420-429) and hand_poses.json (:328-335,486-489)."""
    import json
    from pdfnet_amd.trains.base_trainer import finish_evaluation, write_h2o_scores, write_hand_poses_json
    acc = torch.arange(1.0, 12.0, dtype=torch.float64) / 1000
    acc[10] = 4.0
    ev = finish_evaluation(acc)
    p = str(tmp_path / "H2O-val.txt")
    write_h2o_scores(p, ev)
    write_h2o_scores(p, ev)                                  # appended, like the reference's open(..., 'a')
    lines = open(p).read().splitlines()
    assert len(lines) == 18 and lines[0] == 'eval ' and lines[9] == 'eval '
    assert [l.split(':')[0] for l in lines[1:9]] == ['abs_left_joints_loss_all', 'abs_right_joints_loss_all', 'abs_left_verts_loss_all',
                                                     'abs_right_verts_loss_all', 'off_left_joints_loss_all', 'off_right_joints_loss_all',
                                                     'off_left_verts_loss_all', 'off_right_verts_loss_all']
    assert lines[1] == 'abs_left_joints_loss_all: %.2f' % (1.0 / 4) and lines[4] == 'abs_right_verts_loss_all: %.2f' % (4.0 / 4)
    rows = torch.zeros(3, 128, dtype=torch.float64)
    rows[0, :2] = torch.tensor([2.0, 7.0])
    rows[1, :2] = torch.tensor([1.0, 0.0])
    rows[2, :2] = torch.tensor([1.0, 12.0])
    rows[:, 2:] = torch.arange(3 * 126, dtype=torch.float64).reshape(3, 126)
    q = str(tmp_path / "hand_poses.json")
    write_hand_poses_json(q, rows)
    d = json.load(open(q))
    assert list(d) == ['modality', '1', '2'] and d['modality'] == 'RGBD'
    assert list(d['1']) == ['000000.txt', '000012.txt'] and list(d['2']) == ['000007.txt']
    assert d['2']['000007.txt'] == [float(i) for i in range(126)] and len(d['1']['000012.txt']) == 126


def test_header_is_plain_c_and_a_c_client_links(tmp_path):
    """include/pdfnet_hip.h compiles as C99 with -Wall -Werror and a client without Python / torch links against the library
    (tests/c_abi/c_client.c; the GPU suite runs it)."""
    from tests.util import build_c_client
    exe = build_c_client(tmp_path)
    assert os.path.exists(exe)


def test_bench_gpus_n_launches_its_own_ranks_as_a_child_process():
    """`python bench.py --gpus N` started directly (the driver's form): the parent builds the torch.distributed.run command
    (reference: scripts/train.sh:21, main.py:69-71), runs it as a CHILD and relays its return code; it initialises no device.
    On a box with fewer GPUs it exits non-zero with a one-line reason instead of an AssertionError."""
    import subprocess
    import sys
    import types
    import bench
    seen = {}

    def fake_run(cmd, env=None):
        seen['cmd'], seen['env'] = cmd, env
        return types.SimpleNamespace(returncode=7)
    rc = bench.launch_ranks(8, ['--gpus', '8', '--steps', '3', '--warmup', '1'], run=fake_run, device_count=8)
    assert rc == 7                                               # the child's return code is relayed
    cmd = seen['cmd']
    assert cmd[:3] == [sys.executable, '-m', 'torch.distributed.run'] and '--nnodes=1' in cmd
    assert cmd[cmd.index('--nproc-per-node') + 1] == '8' and cmd[cmd.index('--master-addr') + 1] == '127.0.0.1'
    assert cmd[-7:] == [os.path.join(ROOT, 'bench.py'), '--gpus', '8', '--steps', '3', '--warmup', '1']
    assert seen['env']['HSA_ENABLE_IPC_MODE_LEGACY'] == '0'
    assert not torch.cuda.is_initialized()
    # the real thing on this GPU-less container: non-zero, one line, no traceback
    env = dict(os.environ)
    env.pop('WORLD_SIZE', None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2'], capture_output=True, text=True, env=env, timeout=300)
    if torch.cuda.device_count() < 2:
        assert r.returncode != 0 and 'Traceback' not in r.stderr and 'AssertionError' not in r.stderr
        assert r.stderr.strip().splitlines()[-1].startswith('bench.py: --gpus 2 asked for')
        assert r.stdout.strip() == ''
    # a world size that contradicts --gpus is refused with a reason too
    env['WORLD_SIZE'], env['RANK'], env['LOCAL_RANK'] = '1', '0', '0'
    if not torch.cuda.is_available():
        return
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2'], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0 and 'WORLD_SIZE=1' in r.stderr


def test_rejected_call_leaves_no_armed_slot():
    """the pdf_set_* hand-overs are compat wrappers over the explicit PdfCallOpts of the `_x` entry points; a plain
    entry point takes AND CLEARS every slot before it looks at an argument, so a call that is rejected (PDF_E_BADARG) or returns early
    (empty problem) cannot leave a slot armed for an unrelated later call.  Runs without a GPU: both returns happen before any launch."""
    from pdfnet_amd import hip
    L = hip.lib()
    c = L.cdll
    assert c.pdf_debug_callopts_size() == ctypes.sizeof(hip.CallOpts)

    def arm():
        L.pdf_set_bf16_operands(0x1000, 0x2000)
        L.pdf_set_bf16_output(0x3000)
        L.pdf_set_bn_input_bf16(0x4000)
        L.pdf_set_stats_output(0x5000, 64)
        L.pdf_set_bn_tile_stats(0x6000, 1, 64)
        L.pdf_set_input_affine_relu(0x7000, 0x8000)
        assert L.pdf_debug_armed_slots() == 8
    arm()
    # 9 x 9 taps > MAX_TAPS: rejected
    rc = c.pdf_conv2d_fwd(None, None, None, None, 1, 8, 8, 16, 16, 16, 9, 9, 1, 4, 8, 8, 16, 0, None)
    assert rc == -1 and L.pdf_debug_armed_slots() == 0
    assert L.pdf_stats_result_tiles() == 0
    arm()
    # misaligned leading dimension: pdf_bn_relu_maxk_fwd used to return BEFORE it took its slot
    rc = c.pdf_bn_relu_maxk_fwd(0x10, 5, 6, 4, 4, None, None, None, None, 0.1, 1e-5, 1, 0x20, 6, 0x30, None, None, 0x40, 0x50, None, None)
    assert rc == -1 and L.pdf_debug_armed_slots() == 0
    arm()
    rc = c.pdf_bn_train_fwd(None, 16, 16, 0, None, None, None, None, 0.1, 1e-5, None, 16, 0, None, 16, None, None, None, None, None, None)
    assert rc == 0 and L.pdf_debug_armed_slots() == 0          # empty problem: early return, slots cleared all the same
    # the explicit form reads its options from its argument and never from the thread
    arm()
    o = hip.CallOpts(stats_out=0x5000, stats_cap=64)
    rc = c.pdf_conv2d_fwd_x(None, None, None, None, 1, 8, 8, 16, 16, 16, 9, 9, 1, 4, 8, 8, 16, 0, None, ctypes.byref(o))
    assert rc == -1 and o.stats_tiles == 0 and L.pdf_debug_armed_slots() == 8      # (an `_x` call does not touch the compat slots)
    c.pdf_linear_fwd_pair(None, None, None, None, None, None, 0, 0, 0, 0, 0, 0, 0, None)                 # any plain GEMM-family call clears them
    assert L.pdf_debug_armed_slots() == 0


def test_bf16_storage_defaults_to_the_batches_where_it_wins(monkeypatch):
    """PDFNET_BF16_STORAGE unset = 'auto': bf16 storage of the conv -> BatchNorm tensors for convolutions over >= 32 images (B=64 per GPU:
    +2 %; B=32: +1.7 %, +3.8 % with the epilogue statistics since that step is GPU-bound (round 5); the launch-bound small batches lose),
    never outside bf16 mode; 1 / 0 force it."""
    from pdfnet_amd import functional as F
    monkeypatch.setattr(F, '_GEMM_BF16', True)
    monkeypatch.setattr(F, 'BF16_SHADOWS', True)
    monkeypatch.setattr(F, 'BF16_STORAGE', 'auto')
    assert not F.storage_on(8) and not F.storage_on(31) and F.storage_on(32) and F.storage_on(64) and not F.storage_on()
    monkeypatch.setattr(F, 'BF16_STORAGE', True)
    assert F.storage_on(2) and F.storage_on()
    monkeypatch.setattr(F, 'BF16_STORAGE', False)
    assert not F.storage_on(64)
    # the same rule for the BatchNorm statistics out of the bf16 GEMM epilogue (fp32 mode: always, it is a pure gain there)
    monkeypatch.setattr(F, 'BN_EPILOGUE_STATS_BF16', 'auto')
    assert F._stats_request(True, 1024, 64, 'cpu', batch=16) is None and F._stats_request(True, 1024, 64, 'cpu') is None
    assert F._stats_request(True, 1024, 64, 'cpu', batch=64) is not None
    monkeypatch.setattr(F, 'BN_EPILOGUE_STATS_BF16', True)
    assert F._stats_request(True, 1024, 64, 'cpu', batch=2) is not None
    monkeypatch.setattr(F, 'BF16_STORAGE', 'auto')
    monkeypatch.setattr(F, 'BN_EPILOGUE_STATS_BF16', 'auto')
    monkeypatch.setattr(F, '_GEMM_BF16', False)
    assert not F.storage_on(64)
    assert F._stats_request(True, 1024, 64, 'cpu', batch=2) is not None


def test_resolved_bf16_modes_are_reportable(monkeypatch):
    """the bf16 'auto' rules switch at 48 images per GPU, so B=32 and B=64 follow different rounding models -- the Trainer logs
    what they resolved to (Trainer.resolved_modes); this is the resolution itself."""
    from pdfnet_amd import functional as F
    assert F.bf16_modes(32) == {'gemm': 'fp32'}
    monkeypatch.setattr(F, '_GEMM_BF16', True)
    monkeypatch.setattr(F, 'BF16_SHADOWS', True)
    monkeypatch.setattr(F, 'BF16_STORAGE', 'auto')
    monkeypatch.setattr(F, 'BN_EPILOGUE_STATS_BF16', 'auto')
    lo, hi = F.bf16_modes(16), F.bf16_modes(64)
    assert lo['conv_to_bn_storage'] == 'fp32' and hi['conv_to_bn_storage'] == 'bf16'
    assert 'own pass' in lo['bn_statistics'] and 'epilogue' in hi['bn_statistics']
    assert lo['auto_threshold_batch'] == F.BF16_STORAGE_MIN_BATCH


def test_committed_pmc_profile_matches_the_committed_kernel_sources():
    """bench.py looks `roofline.traffic` (HBM bytes per launch from the rocprofv3 --pmc passes) up in the newest profiles/*_pmc_traffic.json
    and WITHHOLDS it when the kernel sources have changed since the profile was taken (sha256 recorded in the profile).  A kernel edit
    without a re-profile (tools/run_final.sh) would therefore ship a bench line whose `traffic` is null: caught here, without a GPU."""
    import bench
    per_symbol, family, src = bench.pmc_traffic()
    assert per_symbol is not None and family, src
    assert any(k.startswith('wgemm_tn_dma') for k in per_symbol), sorted(per_symbol)[:5]
    hbm, hsrc = bench.pmc_traffic_hbm()
    assert hbm, hsrc
    b16, bsrc = bench.pmc_traffic_bf16()
    assert b16 and any(k.startswith('igemm_bf16_dma') for k in b16), bsrc


def test_committed_bench_line_keeps_the_drivers_contract():
    """The newest committed headline line (profiles/rNN_bench_B32_1gpu.json, written by `python bench.py` on the GPU box) carries every key the
    driver and the judge read, in the types they read them as, and the roofline arithmetic is self-consistent."""
    import glob
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    newest = sorted(glob.glob(os.path.join(root, "profiles", "r[0-9][0-9]_bench_B32_1gpu.json")))[-1]
    d = json.loads(open(newest).read().strip().splitlines()[-1])
    base = json.load(open(os.path.join(root, "BASELINE.json")))
    assert d["metric"].replace("x", "×") in base["metric"], (d["metric"], base["metric"])
    assert d["unit"] == "images/s" and d["higher_is_better"] is True and d["scaling"] == "weak" and d["data"] == "synthetic"
    assert d["n_gpus"] == 1 and d["steps"] >= 1 and d["warmup"] >= 1 and d["vs_baseline"] is None       # (BASELINE.md publishes no number for this metric)
    assert d["dtype"] == "f32" and "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - 32 * 1e3 / d["ms_per_step"]) <= 0.01 * d["value"]                           # whole-job images/s = batch / step time
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s") and r["peak"] == 157.3
    assert abs(r["frac"] - r["achieved"] / r["peak"]) <= 1e-3 and (r["traffic"] is None or r["traffic"] > 0)
    assert 0 < r["step_level"]["executed_frac"] <= r["step_level"]["algorithmic_equivalent_frac"] <= 1.0
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["unit"] == "images/s" and c["cores"] >= 1 and c["value"] > 0 and c["sample"]
    a = d["arithmetic"]                                                                                 # x3 is disclosed, with the all-native step beside it
    assert 1 <= a["x3_mode"] <= 7 and a["native_fp32_mfma_step"]["images_per_s"] > 0 and "native fp32 MFMA" in a["everything_else"]
