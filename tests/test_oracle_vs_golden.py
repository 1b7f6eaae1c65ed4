"""CPU: the oracle restatement (oracle/pdfnet_cpu.py) against fixtures generated from the
reference itself (oracle/make_goldens.py).  This is what pins the oracle."""
import numpy as np
import pytest
import torch

from oracle import pdfnet_cpu as O
from oracle import synth
from tests.util import gold, T, make_opt, pack_outputs, check_packed, surrogate_loss


def test_group_points_level1():
    g = gold("op_group_points_l1")
    pts = T(g["points"])
    idx = O.knn_ball_indices(pts, 512, 64, float(g["r2"]))
    assert (np.sort(idx.numpy(), -1) == g["idx_sorted"]).all()          # bit-exact index sets
    x, ctr = O.group_level1(pts, 512, 64, float(g["r2"]))
    assert np.allclose(x.sum(-1).numpy(), g["grouped_sum"], atol=1e-5)
    assert np.array_equal(x.abs().amax(-1).numpy(), g["grouped_absmax"])
    assert np.array_equal(ctr.numpy(), g["centers"])


def test_group_points_level2():
    g = gold("op_group_points_l2")
    f = T(g["feat"])
    idx = O.knn_ball_indices(f[:, :3].transpose(1, 2), 128, 64, float(g["r2"]))
    assert (np.sort(idx.numpy(), -1) == g["idx_sorted"]).all()
    x, ctr = O.group_level2(f, 128, 64, float(g["r2"]))
    assert np.allclose(x.sum(-1).numpy(), g["grouped_sum"], atol=1e-4)
    assert np.array_equal(x.abs().amax(-1).numpy(), g["grouped_absmax"])
    assert np.array_equal(ctr.numpy(), g["centers"])


def test_gather_sft_l2norm_nms():
    g = gold("op_gather_feat")
    assert np.array_equal(O.gather_hw(T(g["feat"]), T(g["ind"])).numpy(), g["out"])
    g = gold("op_sft")
    m = O.SFTLayer(13, 6)
    m.load_state_dict({k[2:]: T(v) for k, v in g.items() if k.startswith("w_")})
    assert np.allclose(m(T(g["fea"]), T(g["cond"])).detach().numpy(), g["out"], atol=1e-6)
    g = gold("op_l2norm")
    m = O.L2Norm(5, 10)
    m.weight.data = T(g["weight"])
    assert np.allclose(m(T(g["x"])).detach().numpy(), g["out"], atol=1e-6)
    g = gold("op_nms_topk")
    assert np.array_equal(O.nms_topk_center(T(g["hm"])).numpy(), g["ind"])


@pytest.mark.parametrize("V,Fd", [(63, 16), (126, 8), (252, 8)])
def test_gcn_block(V, Fd):
    g = gold("op_gcn_block_V%d" % V)
    L = O.load_graph_constants()['L_left'][{63: 0, 126: 1, 252: 2}[V]]
    blk = O.GCNResBlock(Fd, Fd // 2, L, 0.0)
    blk.load_state_dict({k[2:]: T(v) for k, v in g.items() if k.startswith("w_")})
    x = T(g["x"])
    assert np.allclose(O.cheby_k2(x, blk.graph_L, blk.fc1).detach().numpy(), g["cheby_fc1"], atol=2e-5)
    assert np.allclose(blk(x).detach().numpy(), g["out"], atol=2e-5)


def test_attention_blocks():
    g = gold("op_self_attn")
    sa = O.SelfAttn(16, 4, 0.0)
    sa.load_state_dict({k[2:]: T(v) for k, v in g.items() if k.startswith("w_")})
    assert np.allclose(sa(T(g["x"])).detach().numpy(), g["out"], atol=2e-5)
    g = gold("op_inter_attn")
    ia = O.InterAttn(16, 4, 0.0)
    ia.load_state_dict({k[2:]: T(v) for k, v in g.items() if k.startswith("w_")})
    oL, oR = ia(T(g["x"]), T(g["y"]))
    assert np.allclose(oL.detach().numpy(), g["outL"], atol=2e-5)
    assert np.allclose(oR.detach().numpy(), g["outR"], atol=2e-5)


@pytest.mark.parametrize("level", [0, 1, 2])
def test_dualgraph_layer_at_the_real_dimensions(level):
    """The oracle's DualGraphLayer against the reference's (DualGraph.py:21-92) on op_dualgraph_layer_L*: output, input gradient, gradient norms."""
    from oracle import synth
    g = gold("op_dualgraph_layer_L%d" % level)
    V, cin, cout = synth.DUALGRAPH_DIMS[level]
    gc = O.load_graph_constants()
    layer = O.DualGraphLayer(V, cin, cout, gc['L_left'][level], gc['L_right'][level], 4, [12, 24, 48][level], 256, 6, (256, 128, 64)[level], 4, 0.0)
    layer.load_state_dict(synth.det_state_dict(layer.state_dict(), salt=level + 1))
    xn, gyn = synth.dualgraph_case(level)
    x = T(xn).requires_grad_()
    Lf, Rf = layer(x[0], x[1])
    out = torch.stack((Lf, Rf))
    out.backward(T(gyn))
    assert np.abs(out.detach().numpy() - g["out"]).max() <= 2e-5 * max(1.0, np.abs(g["out"]).max())
    assert np.abs(x.grad[..., level::8].numpy() - g["dx_sub"]).max() <= 1e-4 * np.abs(g["dx_sub"]).max()
    grads = {n: p.grad for n, p in layer.named_parameters() if p.grad is not None}
    assert set(grads) == set(str(n) for n in g["grad_names"])
    for i, n in enumerate(g["grad_names"]):
        if str(n).endswith("w_ks.bias"):
            continue
        assert abs(float(grads[str(n)].double().norm()) / float(g["grad_norm"][i]) - 1) <= 1e-3, n


def test_mano_layer():
    g = gold("op_mano_layer")
    for side in ("left", "right"):
        c = synth.synthetic_mano_consts(side)
        v, j = O.mano_lbs(c, T(g[side + "_rot"]), T(g[side + "_pose"]), T(g[side + "_shape"]),
                          trans=T(g[side + "_trans"]), side=side)
        assert np.allclose(v.numpy(), g[side + "_verts"], atol=2e-6)
        assert np.allclose(j.numpy(), g[side + "_joints"], atol=2e-6)
        reg = O.full_regressor(c['J_regressor'])
        assert np.allclose(O.regress_joints(reg, v).numpy(), g[side + "_full_regressor_joints"], atol=2e-6)


@pytest.fixture(scope="module")
def model_and_batch():
    torch.set_num_threads(8)
    m = O.load_model_cpu(make_opt(256))
    sd = synth.det_state_dict(m.state_dict())
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    b = synth.to_torch(synth.synthetic_batch(2, 256, seed=1, variant='mixed'))
    return m, sd, b


def test_state_dict_manifest(model_and_batch, golden_dir):
    m, sd, _ = model_and_batch
    import hashlib
    import os
    lines = open(os.path.join(golden_dir, "state_dict_manifest.txt")).read().strip().split("\n")
    names = [l.split()[0] for l in lines if not l.startswith("#")]
    assert names == list(m.state_dict().keys()) and len(names) == 1287
    h = hashlib.sha256()
    for k, v in sd.items():
        h.update(k.encode())
        h.update(v.numpy().tobytes())
    assert lines[-1].split()[-1] == h.hexdigest()          # generator reproduces the same bytes


@pytest.mark.slow
def test_e2e_eval(model_and_batch):
    m, sd, b = model_and_batch
    g = gold("e2e_eval_B2_R256")
    m.load_state_dict(sd)
    m.eval()
    with torch.no_grad():
        res = m(b['input'], b['choose'], b['cloud'], b['depth'], b['ind'], b['K_new'], b['valid'])
        check_packed(pack_outputs(res, b['ind']), g)
        res2 = m(b['input'], b['choose'], b['cloud'], b['depth'], None, b['K_new'], b['valid'])
    assert np.array_equal(res2[3]['ind'].numpy(), g["pred_ind"])          # predicted centres bit-exact
    assert np.allclose(res2[0]['verts3d']['left'].numpy(), g["pred_ind_verts3d_left"], atol=1e-4)


@pytest.mark.slow
def test_e2e_train_and_grads(model_and_batch):
    m, sd, b = model_and_batch
    g = gold("e2e_train_B2_R256")
    m.load_state_dict(sd)
    m.train()
    m.zero_grad()
    res = m(b['input'], b['choose'], b['cloud'], b['depth'], b['ind'], b['K_new'], b['valid'])
    check_packed(pack_outputs(res, b['ind']), g, abs_tol=1e-3, rel_tol=1e-4)
    loss = surrogate_loss(res)
    assert abs(loss.item() - float(g["loss"][0])) < 1e-4 * abs(float(g["loss"][0]))
    loss.backward()
    named = dict(m.named_parameters())
    for k, v in g.items():
        if k.startswith("gradnorm::"):
            n = named[k[10:]].grad.double().norm().item()
            assert abs(n - float(v[0])) <= 1e-3 * float(v[0]) + 1e-9, (k, n, v)
    assert sum(p.grad is None for p in named.values()) == int(g["n_params_without_grad"][0])
    new = m.state_dict()
    for k, v in g.items():
        if k.startswith("stat::"):
            assert np.allclose(new[k[6:]].numpy(), v, atol=1e-5)


def test_depth2pcl_front_end():
    """oracle depth2pcl == the reference's (intaghand_encoder.py:369-491) under the same numpy seed, bit for bit."""
    g = gold("op_depth2pcl")
    for name, valid in (("both", [1, 1]), ("left_only", [1, 0])):
        np.random.seed(0)
        ch, cl = O.depth2pcl(g["depth"], g["mask"], g["K"], np.array(valid))
        assert np.array_equal(ch, g["choose_" + name])
        assert np.array_equal(cl, g["cloud_" + name])
    # the fixture exercises both selection paths
    _, cl_ = O.depth_candidates(g["depth"], g["mask"][1], g["K"])
    _, cr_ = O.depth_candidates(g["depth"], g["mask"][0], g["K"])
    assert 10 <= len(cl_) <= 1024 < len(cr_)


def test_fps_matches_reference_helper():
    """oracle.fps_order (picks in order) against the reference's farthest_point_sampling_fast (np.unique of the picks)."""
    from oracle import pdfnet_cpu as O
    g = gold("op_fps")
    for name in ("a", "b", "dup"):
        order = O.fps_order(g["pts_" + name], int(g["S_" + name][0]), int(g["start_" + name][0]))
        assert np.array_equal(np.unique(order), g["unique_" + name]), name



def _loss_fixture_inputs(B=3, R=256):
    from oracle import loss_cpu as LC
    from pdfnet_amd.synthetic import synthetic_loss_constants, synthetic_train_batch
    from tests.util import synthetic_model_outputs, ROOT
    import os
    consts = synthetic_loss_constants()
    z = np.load(os.path.join(ROOT, "pdfnet_amd", "data", "gcn_core.npz"))
    conv = {h: LC.Converter(z['graph_perm_' + h], z['graph_perm_reverse_' + h]) for h in ('left', 'right')}
    batch = synthetic_train_batch(B, R, seed=5, consts=consts)
    batch['valid'][1, 1] = 0.0

    def outputs():
        result, params, hand, other = synthetic_model_outputs(B, R, 9)
        other['converter_left'], other['converter_right'] = conv['left'], conv['right']
        return result, params, hand, other
    opt = make_opt(R, size_train=[R, R], down_ratio=4, center_weight=200.0, reproj_weight=1.0, bone_dir_weight=200.0)
    return opt, consts, batch, outputs


def test_loss_oracle_matches_reference_golden():
    """oracle/loss_cpu.py against the reference's own CtdetLoss.forward (fixture written by oracle/make_goldens.py
    loss_golden): per-sample loss, all 16 statistics at alpha = 0 and 1, and the test-mode 9-tuple."""
    from oracle import loss_cpu as LC
    g = gold("loss_ctdet_B3_R256")
    opt, consts, batch, outputs = _loss_fixture_inputs()
    for epoch in (0, 25):
        loss, stats = LC.ctdet_loss(opt, consts, *outputs(), batch, 'train', epoch)
        assert np.allclose(loss.numpy(), g["loss_e%d" % epoch], rtol=1e-6, atol=1e-5), epoch
        n = 0
        for k, v in g.items():
            if k.startswith("stat_e%d::" % epoch):
                got = torch.as_tensor(stats[k.split("::")[1]]).reshape(-1).numpy()
                assert np.allclose(got, v, rtol=1e-6, atol=1e-7), (epoch, k)
                n += 1
        assert n == 16
    tup = LC.ctdet_loss(opt, consts, *outputs(), batch, 'test', 0)
    assert len(tup) == 9
    for i, t in enumerate(tup):
        assert np.allclose(t.numpy(), g["test%d" % i], rtol=1e-6, atol=1e-6), i
    # evaluation metric (base_trainer.py:244-323): the formula on the golden tuple itself
    m = LC.evaluation_metrics(tuple(T(g["test%d" % i]) for i in range(9)), (batch['lms_left_gt'], batch['lms_right_gt']))
    ref = np.linalg.norm(g["test1"][:, 0] - g["test3"][:, 0], axis=-1).mean() * 1000
    assert abs(m['abs_left_joints'] - ref) < 1e-3 * ref


def test_demo_asset_pair_config1():
    """BASELINE config 1: the reference's demo forward on its own asset pair assets/H2O/{color,depth}/000002.png
    (fixture = network input + the reference's outputs, oracle/make_demo_golden.py).  The CPU oracle on the same input and
    the clouds the reference's depth2pcl drew must reproduce the reference's outputs; its own front end must draw from the
    same candidate set."""
    from tests.util import demo_fixture_inputs, demo_state_dict, pack_demo
    g, b = demo_fixture_inputs()
    o = O.load_model_cpu(make_opt(256))
    o.load_state_dict(demo_state_dict(o.state_dict(), g))
    o.eval()
    with torch.no_grad():
        res = o(b['input'], b['choose'], b['cloud'], b['depth'], None, b['K_new'], b['valid'])
    assert np.array_equal(res[3]['ind'].numpy(), g["pred_ind"])                     # centres picked from the heat-map: bit-exact
    check_packed(pack_demo(res), g, abs_tol=1e-4, rel_tol=1e-5)
    mask = res[3]['mask']
    assert [(mask[0, c] > 0.5).sum().item() for c in range(2)] == g["mask_pos_count"].tolist()
    # the clouds are real depth: 1024 distinct pixels per hand, each inside the predicted mask and the +-8 cm window
    for hi, ch in ((0, 1), (1, 0)):                                                # hand 0 = left = mask channel 1 (:376-377)
        pix = g["choose"][hi]
        assert len(np.unique(pix)) == 1024
        assert (mask[0, ch].flatten()[torch.from_numpy(pix)] > 0.5).all()
        z = g["cloud"][hi][:, 2]
        assert z.max() - z.min() < 0.16 + 1e-6 and z.min() > 0.2
