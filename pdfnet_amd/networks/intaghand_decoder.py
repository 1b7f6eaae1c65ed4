"""MI355X-native counterpart of lib/models/networks/intaghand_decoder.py + model_attn/{DualGraph,gcn,
self_attn,inter_attn,img_attn}.py: the IntagHand-style dual-hand GCN / attention mesh decoder.

Same module tree and state_dict keys as the reference `decoder` (intaghand_decoder.py:75-242).  The graph
Laplacians are fixed-width ELL tables (<= 11 non-zeros per row) instead of the reference's densified
matrices (gcn.py:79-86); features stay row-major [B, V, F] end to end (no permutes).
"""
import os

import numpy as np
import torch
import torch.nn as nn

from .. import functional as F
from .layers import Conv2d, Embedding, LayerNorm, Linear, xavier_

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "data", "gcn_core.npz")
IMG_SIZE = 384          # intaghand_decoder.py:17 -- scales verts2d whatever the input resolution (:225-227)


def _csr_to_ell(indptr, indices, data, V):
    width = int(np.diff(indptr).max())
    col = np.zeros((V, width), np.int32)
    val = np.zeros((V, width), np.float32)
    for v in range(V):
        a, b = indptr[v], indptr[v + 1]
        col[v, :b - a] = indices[a:b]
        val[v, :b - a] = data[a:b]
    return col, val


def load_graph_constants():
    """Data asset built by tools/convert_gcn_core.py from the reference's gcn_core/*.pkl."""
    z = np.load(_DATA)
    out = {'upsample': torch.from_numpy(z['upsample']), 'dense_coor': torch.from_numpy(z['dense_coor'])}
    for hand in ('left', 'right'):
        ells = []
        for V in (63, 126, 252):
            ip, ix, dt = z['L_%s_%d_indptr' % (hand, V)], z['L_%s_%d_indices' % (hand, V)], z['L_%s_%d_data' % (hand, V)]
            col, val = _csr_to_ell(ip, ix, dt, V)
            # transpose (for the backward): CSR of L^T
            rows = np.repeat(np.arange(V), np.diff(ip))
            order = np.lexsort((rows, ix))
            tptr = np.zeros(V + 1, np.int64)
            np.add.at(tptr, ix + 1, 1)
            tptr = np.cumsum(tptr)
            colT, valT = _csr_to_ell(tptr, rows[order], dt[order], V)
            ells.append(tuple(torch.from_numpy(a) for a in (col, val, colT, valT)))
        out['ell_' + hand] = ells
        out['perm_' + hand] = torch.from_numpy(z['graph_perm_' + hand])
        out['perm_rev_' + hand] = torch.from_numpy(z['graph_perm_reverse_' + hand][:778])
    for i in range(3):                                                  # paired launches want one ELL width per level
        l, r = out['ell_left'][i], out['ell_right'][i]
        padded = [[], []]
        for a, b in zip(l, r):
            w = max(a.shape[1], b.shape[1])
            for k, t in enumerate((a, b)):
                padded[k].append(torch.nn.functional.pad(t, (0, w - t.shape[1])))
        out['ell_left'][i], out['ell_right'][i] = tuple(padded[0]), tuple(padded[1])
    return out


class GCN_vert_convert:
    """intaghand_decoder.py:32-43.  The index tables are (non-persistent) buffers of the decoder module so they
    move with .cuda()/.to() and no host->device copy happens inside a captured step."""

    def __init__(self, owner, hand):
        self._owner, self._hand = owner, hand

    @property
    def graph_perm(self):
        return getattr(self._owner, 'graph_perm_' + self._hand)

    @property
    def graph_perm_reverse(self):
        return getattr(self._owner, 'graph_perm_reverse_' + self._hand)

    def vert_to_GCN(self, x):
        return x[:, self.graph_perm]

    def GCN_to_vert(self, x):
        return x[:, self.graph_perm_reverse]


class GCN_ResBlock(nn.Module):
    """gcn.py:72-110.  norm1's output is discarded by the reference (:103-104); it is kept as parameters only."""

    def __init__(self, cin, cout, ell, drop):
        super().__init__()
        for name, t in zip(('ell_col', 'ell_val', 'ell_colT', 'ell_valT'), ell):
            self.register_buffer(name, t.clone(), persistent=False)
        self.norm1 = LayerNorm(cin)
        self.fc1 = Linear(cin * 2, cout)
        self.norm2 = LayerNorm(cout)
        self.fc2 = Linear(cout * 2, cout)
        self.shortcut = Linear(cin, cout)
        self.norm3 = LayerNorm(cout)
        self.p = drop

    @property
    def ell(self):
        return (self.ell_col, self.ell_val, self.ell_colT, self.ell_valT)


def _lin2(l, r, x, act=F.ACT_NONE):
    return F.linear_pair(x, l.weight, l.bias, r.weight, r.bias, act)


def _ln2(l, r, x, act=F.ACT_NONE, add=None, p=0.0, training=False):
    return F.layer_norm_fused(x, l.weight, l.bias, l.eps, act, add, p, training, r.weight, r.bias)


def gcn_block_pair(bl, br, x, relu_out):
    """Both hands' GCN_ResBlock (gcn.py:99-110) on x [2,B,V,C]: 7 launches --
    cheby, fc1 | LN2+ReLU | cheby, fc2 | shortcut | dropout + add + LN3 [+ the ReLU GraphLayer puts between blocks]."""
    y = _lin2(bl.fc1, br.fc1, F.cheby2_pair(x, bl.ell, br.ell))
    h = _ln2(bl.norm2, br.norm2, y, F.ACT_RELU)
    y = _lin2(bl.fc2, br.fc2, F.cheby2_pair(h, bl.ell, br.ell))
    s = _lin2(bl.shortcut, br.shortcut, x)
    _, out = _ln2(bl.norm3, br.norm3, s, F.ACT_RELU if relu_out else F.ACT_NONE, add=y, p=bl.p, training=bl.training)
    return out


class GraphLayer(nn.Module):
    def __init__(self, cin, cout, ell, n, drop):
        super().__init__()
        self.GCN_blocks = nn.ModuleList([GCN_ResBlock(cin if i == 0 else cout, cout, ell, drop) for i in range(n)])
        xavier_(self)


class MLP_res_block(nn.Module):
    """self_attn.py:17-33."""

    def __init__(self, d, hid, drop):
        super().__init__()
        self.layer_norm = LayerNorm(d)
        self.fc1, self.fc2 = Linear(d, hid), Linear(hid, d)
        self.p = drop


def attn_tail_pair(fl, fr, x, o, p, training):
    """x + dropout(o), then both hands' MLP_res_block (self_attn.py:24-33) on the sum: the residual sum and the block's
    LayerNorm are one kernel, its closing dropout + residual another."""
    z, hn = _ln2(fl.layer_norm, fr.layer_norm, x, add=o, p=p, training=training)
    h = F.dropout(_lin2(fl.fc1, fr.fc1, hn, F.ACT_RELU), fl.p, training)
    return F.dropout_add(_lin2(fl.fc2, fr.fc2, h), z, fl.p, training)


class SelfAttn(nn.Module):
    """self_attn.py:36-85."""

    def __init__(self, d, heads, drop):
        super().__init__()
        self.n_heads, self.p = heads, drop
        self.w_qs, self.w_ks, self.w_vs = Linear(d, d), Linear(d, d), Linear(d, d)
        self.layer_norm = LayerNorm(d)
        self.fc = Linear(d, d)
        self.ff = MLP_res_block(d, d, drop)


def self_attn_pair(al, ar, x):
    """L_self_attn_layer / R_self_attn_layer (inter_attn.py:108-109) on x [2,B,V,d]; attention itself has no parameters,
    so it runs once over the 2B stacked samples."""
    h = _ln2(al.layer_norm, ar.layer_norm, x)
    a = F.attention(_lin2(al.w_qs, ar.w_qs, h), _lin2(al.w_ks, ar.w_ks, h), _lin2(al.w_vs, ar.w_vs, h),
                    al.n_heads, al.p, al.training)
    return attn_tail_pair(al.ff, ar.ff, x, _lin2(al.fc, ar.fc, a), al.p, al.training)


class inter_attn(nn.Module):
    """inter_attn.py:38-125 (w_qs/w_ks/w_vs/fc shared by both directions)."""

    def __init__(self, d, heads, drop):
        super().__init__()
        self.n_heads, self.p = heads, drop
        self.L_self_attn_layer = SelfAttn(d, heads, drop)
        self.R_self_attn_layer = SelfAttn(d, heads, drop)
        self.w_qs, self.w_ks, self.w_vs, self.fc = (Linear(d, d) for _ in range(4))
        self.layer_norm1, self.layer_norm2 = LayerNorm(d), LayerNorm(d)
        self.ffL, self.ffR = MLP_res_block(d, d, drop), MLP_res_block(d, d, drop)
        xavier_(self)

    def forward(self, x):
        """x [2,B,V,d] = (left, right) features."""
        x = self_attn_pair(self.L_self_attn_layer, self.R_self_attn_layer, x)
        n = _ln2(self.layer_norm1, self.layer_norm2, x)
        q, k, v = self.w_qs(n), self.w_ks(n), self.w_vs(n)              # shared projections: one GEMM for both hands
        # left queries attend to right keys / values and vice versa (:82-105): the other hand is B samples away
        c = F.attention(q, k, v, self.n_heads, self.p, self.training, kv_shift=x.shape[1])
        return attn_tail_pair(self.ffL, self.ffR, x, self.fc(c), self.p, self.training)


class img_ex(nn.Module):
    """img_attn.py:95-113: constructed by the reference but never called (DualGraph.py:86-87) -- parameters only,
    kept so checkpoints load unchanged."""

    def __init__(self, img_size, img_c, grid, grid_c, d, heads, drop):
        super().__init__()
        enc = nn.Module()
        enc.position_embeddings = Embedding(grid * grid, grid_c)
        p = img_size // grid
        enc.proj = Conv2d(img_c, grid_c, p, p, 0)
        enc.self_attn = SelfAttn(grid_c, heads, drop)
        self.encoder = enc
        att = nn.Module()
        att.fc = Linear(grid_c, d)
        att.Attn = SelfAttn(d, heads, drop)
        self.attn = att
        xavier_(self)


class DualGraphLayer(nn.Module):
    """DualGraph.py:21-96.  Both hands travel as one stacked tensor [2,B,V,C]; every layer that has per-hand parameters
    runs as a paired launch (F.linear_pair / layer_norm_fused / cheby2_pair)."""

    def __init__(self, V, cin, cout, ellL, ellR, n, img_size, img_c, grid_c, heads, drop):
        super().__init__()
        self.position_embeddings = Embedding(V, cin)
        self.graph_left = GraphLayer(cin, cout, ellL, n, drop)
        self.graph_right = GraphLayer(cin, cout, ellR, n, drop)
        self.img_ex_left = img_ex(img_size, img_c, 6, grid_c, cout, heads, drop)
        self.img_ex_right = img_ex(img_size, img_c, 6, grid_c, cout, heads, drop)
        self.attn = inter_attn(cout, heads, drop)

    def forward(self, x):
        x = x + self.position_embeddings.weight
        if F.mesh_level_ok(self, x):
            return F.mesh_level(self, x)                    # csrc/meshdec.hip: the whole layer in 3 (+ 5 backward) launches, fp32 mode
        blocks = list(zip(self.graph_left.GCN_blocks, self.graph_right.GCN_blocks))
        for i, (bl, br) in enumerate(blocks):
            x = gcn_block_pair(bl, br, x, relu_out=i != len(blocks) - 1)
        return self.attn(x)


class DualGraph(nn.Module):
    def __init__(self, layers):
        super().__init__()
        self.layers = nn.ModuleList(layers)

    def forward(self, x):
        last = len(self.layers) - 1
        for i, layer in enumerate(self.layers):
            x = layer(x)
            if i != last:                                   # nearest x2 on the vertex axis (DualGraph.py:11-18)
                x = x.repeat_interleave(2, dim=2)
        return x


def projection_batch(scale, trans2d, pts, img_size):
    """lib/utils/utils.py:231-249."""
    s = (scale * img_size).view(-1, 1, 1)
    t = (trans2d * img_size / 2 + img_size / 2).unsqueeze(1)
    return s * pts[..., :2] + t


class decoder(nn.Module):
    def __init__(self, opt, dropout=0.05, num_attn_heads=4):
        super().__init__()
        g = load_graph_constants()
        self.register_buffer('dense_coor', g['dense_coor'].clone())
        for h in ('left', 'right'):
            self.register_buffer('graph_perm_' + h, g['perm_' + h].clone(), persistent=False)
            self.register_buffer('graph_perm_reverse_' + h, g['perm_rev_' + h].clone(), persistent=False)
        self.converter = {h: GCN_vert_convert(self, h) for h in ('left', 'right')}
        cin, cout = opt.GCN_IN_DIM, opt.GCN_OUT_DIM
        layers = []
        for i, V in enumerate((63, 126, 252)):
            layers.append(DualGraphLayer(V, cin[i], cout[i], g['ell_left'][i], g['ell_right'][i], opt.graph_layer_num,
                                         [12, 24, 48][i], opt.DECONV_DIMS[i], opt.IMG_DIMS[i], num_attn_heads, dropout))
        self.dual_gcn = DualGraph(layers)
        d0 = cin[0] - 3
        self.gf_layer_left = nn.Sequential(Linear(1024, d0), LayerNorm(d0))
        self.gf_layer_right = nn.Sequential(Linear(1024, d0), LayerNorm(d0))
        self.unsample_layer = Linear(252, 778, bias=False)
        self.unsample_layer.weight.data.copy_(g['upsample'])            # :158-160
        self.coord_head = Linear(cout[-1], 3)
        self.avg_head = Linear(252, 1)
        self.params_head = Linear(cout[-1], 3)
        self.root_head = Linear(cout[-1], 3)
        for m in (self.gf_layer_left, self.gf_layer_right, self.coord_head, self.avg_head, self.params_head, self.root_head):
            xavier_(m)
        self._pe_key, self._pe = None, None

    def get_converter(self):
        return self.converter

    def hand_pe(self):
        """:170-178: dense_coor*2-1 in GCN order (1008 nodes), average-pooled by 16 -> [2,1,63,3] (left, right)."""
        key = (self.dense_coor._version, self.dense_coor.device)
        if self._pe_key != key:
            with torch.no_grad():
                c = (self.dense_coor * 2 - 1).unsqueeze(0)
                self._pe = torch.stack([self.converter[h].vert_to_GCN(c).view(1, 63, 16, 3).mean(2) for h in ('left', 'right')])
            self._pe_key = key
        return self._pe

    def forward(self, gl, gr):
        bs = gl.shape[0]
        g = torch.stack((gl, gr))                                                         # [2,B,1024]
        f = _ln2(self.gf_layer_left[1], self.gf_layer_right[1], _lin2(self.gf_layer_left[0], self.gf_layer_right[0], g))
        x = torch.cat([f.unsqueeze(2).expand(-1, -1, 63, -1), self.hand_pe().expand(-1, bs, -1, -1)], -1)
        x = self.dual_gcn(x)                                                              # [2,B,252,C]
        # the heads are shared by both hands (:203-227): one launch each over the stacked features
        t = self.avg_head(x.transpose(2, 3))[..., 0]                                      # [2,B,C]      (:205)
        p = self.params_head(t)
        rt = self.root_head(t)
        c3 = self.coord_head(x)                                                           # [2,B,252,3]
        up3 = self.unsample_layer(c3.transpose(2, 3)).transpose(2, 3).contiguous()        # [2,B,778,3]
        sc, tr = p[..., 0], p[..., 1:]
        c2 = projection_batch(sc.reshape(-1), tr.reshape(-1, 2), c3.reshape(2 * bs, -1, 3), IMG_SIZE).view(2, bs, -1, 2)
        up2 = projection_batch(sc.reshape(-1), tr.reshape(-1, 2), up3.reshape(2 * bs, -1, 3), IMG_SIZE).view(2, bs, -1, 2)
        hands = ('left', 'right')
        scale = {h: sc[i] for i, h in enumerate(hands)}
        trans2d = {h: tr[i] for i, h in enumerate(hands)}
        root = {h: rt[i] for i, h in enumerate(hands)}
        v3 = {h: c3[i] for i, h in enumerate(hands)}
        v2 = {h: c2[i] for i, h in enumerate(hands)}
        result = {'verts3d': {h: up3[i] for i, h in enumerate(hands)}, 'verts2d': {h: up2[i] for i, h in enumerate(hands)}}
        other = {'verts3d_MANO_list': {'left': [], 'right': []}, 'verts2d_MANO_list': {'left': [], 'right': []}}
        for hand in hands:                                                                # :230-240
            for key, src in (('verts3d_MANO_list', v3), ('verts2d_MANO_list', v2)):
                other[key][hand].append(self.converter[hand].GCN_to_vert(src[hand].repeat_interleave(4, dim=1)))
        return (result, {'scale': scale, 'trans2d': trans2d, 'root': root},
                [{'verts3d': v3, 'verts2d': v2}], other)


def load_decoder(opt, encoder_info=None):
    """intaghand_decoder.py:245-278."""
    return decoder(opt, dropout=0.05, num_attn_heads=4)
