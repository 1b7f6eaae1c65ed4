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

    def forward(self, x):
        ell = (self.ell_col, self.ell_val, self.ell_colT, self.ell_valT)
        y = self.fc1(F.cheby2(x, ell))
        y = self.fc2(F.cheby2(F.relu(self.norm2(y)), ell))
        y = F.dropout(y, self.p, self.training)
        return self.norm3(y + self.shortcut(x))


class GraphLayer(nn.Module):
    def __init__(self, cin, cout, ell, n, drop):
        super().__init__()
        self.GCN_blocks = nn.ModuleList([GCN_ResBlock(cin if i == 0 else cout, cout, ell, drop) for i in range(n)])
        xavier_(self)

    def forward(self, x):
        last = len(self.GCN_blocks) - 1
        for i, blk in enumerate(self.GCN_blocks):
            x = blk(x)
            if i != last:
                x = F.relu(x)
        return x


class MLP_res_block(nn.Module):
    """self_attn.py:17-33."""

    def __init__(self, d, hid, drop):
        super().__init__()
        self.layer_norm = LayerNorm(d)
        self.fc1, self.fc2 = Linear(d, hid), Linear(hid, d)
        self.p = drop

    def forward(self, x):
        h = F.dropout(self.fc1(self.layer_norm(x), F.ACT_RELU), self.p, self.training)
        return x + F.dropout(self.fc2(h), self.p, self.training)


class SelfAttn(nn.Module):
    """self_attn.py:36-85."""

    def __init__(self, d, heads, drop):
        super().__init__()
        self.n_heads, self.p = heads, drop
        self.w_qs, self.w_ks, self.w_vs = Linear(d, d), Linear(d, d), Linear(d, d)
        self.layer_norm = LayerNorm(d)
        self.fc = Linear(d, d)
        self.ff = MLP_res_block(d, d, drop)

    def forward(self, x):
        h = self.layer_norm(x)
        a = F.attention(self.w_qs(h), self.w_ks(h), self.w_vs(h), self.n_heads, self.p, self.training)
        return self.ff(x + F.dropout(self.fc(a), self.p, self.training))


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

    def forward(self, Lf0, Rf0):
        Lf, Rf = F.parallel(lambda: self.L_self_attn_layer(Lf0), lambda: self.R_self_attn_layer(Rf0))
        l2, r2 = self.layer_norm1(Lf), self.layer_norm2(Rf)
        B = l2.shape[0]
        both = torch.cat((l2, r2), 0)                                   # shared projections: one GEMM for both hands
        q, k, v = self.w_qs(both), self.w_ks(both), self.w_vs(both)
        r2l = F.attention(q[:B], k[B:], v[B:], self.n_heads, self.p, self.training)
        l2r = F.attention(q[B:], k[:B], v[:B], self.n_heads, self.p, self.training)
        o = self.fc(torch.cat((r2l, l2r), 0))
        Lf = self.ffL(Lf + F.dropout(o[:B], self.p, self.training))
        Rf = self.ffR(Rf + F.dropout(o[B:], self.p, self.training))
        return Lf, Rf


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
    """DualGraph.py:21-96."""

    def __init__(self, V, cin, cout, ellL, ellR, n, img_size, img_c, grid_c, heads, drop):
        super().__init__()
        self.position_embeddings = Embedding(V, cin)
        self.graph_left = GraphLayer(cin, cout, ellL, n, drop)
        self.graph_right = GraphLayer(cin, cout, ellR, n, drop)
        self.img_ex_left = img_ex(img_size, img_c, 6, grid_c, cout, heads, drop)
        self.img_ex_right = img_ex(img_size, img_c, 6, grid_c, cout, heads, drop)
        self.attn = inter_attn(cout, heads, drop)

    def forward(self, Lf, Rf):
        pe = self.position_embeddings.weight.unsqueeze(0)
        Lg, Rg = F.parallel(lambda: self.graph_left(Lf + pe), lambda: self.graph_right(Rf + pe))
        return self.attn(Lg, Rg)


class DualGraph(nn.Module):
    def __init__(self, layers):
        super().__init__()
        self.layers = nn.ModuleList(layers)

    def forward(self, Lf, Rf):
        last = len(self.layers) - 1
        for i, layer in enumerate(self.layers):
            Lf, Rf = layer(Lf, Rf)
            if i != last:                                   # nearest x2 on the vertex axis (DualGraph.py:11-18)
                Lf, Rf = Lf.repeat_interleave(2, dim=1), Rf.repeat_interleave(2, dim=1)
        return Lf, Rf


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

    def get_converter(self):
        return self.converter

    def hand_pe(self, bs, hand):
        """:170-178: dense_coor*2-1 in GCN order (1008 nodes), average-pooled by 16 -> [bs,63,3]."""
        pe = self.converter[hand].vert_to_GCN((self.dense_coor * 2 - 1).unsqueeze(0))
        return pe.view(1, 63, 16, 3).mean(2).expand(bs, -1, -1)

    def forward(self, gl, gr):
        bs = gl.shape[0]
        feats = []
        for hand, gf, layer in (('left', gl, self.gf_layer_left), ('right', gr, self.gf_layer_right)):
            f = layer[1](layer[0](gf))
            feats.append(torch.cat([f.unsqueeze(1).expand(-1, 63, -1), self.hand_pe(bs, hand)], -1))
        Lf, Rf = self.dual_gcn(feats[0], feats[1])
        scale, trans2d, root, v3, v2 = {}, {}, {}, {}, {}
        result = {'verts3d': {}, 'verts2d': {}}
        for hand, f in (('left', Lf), ('right', Rf)):
            t = self.avg_head(f.transpose(1, 2))[..., 0]                                 # [B,64]      (:205)
            p = self.params_head(t)
            scale[hand], trans2d[hand], root[hand] = p[:, 0], p[:, 1:], self.root_head(t)
            v3[hand] = self.coord_head(f)                                                # [B,252,3]
            v2[hand] = projection_batch(scale[hand], trans2d[hand], v3[hand], IMG_SIZE)
            result['verts3d'][hand] = self.unsample_layer(v3[hand].transpose(1, 2)).transpose(1, 2)
            result['verts2d'][hand] = projection_batch(scale[hand], trans2d[hand], result['verts3d'][hand], IMG_SIZE)
        other = {'verts3d_MANO_list': {'left': [], 'right': []}, 'verts2d_MANO_list': {'left': [], 'right': []}}
        for hand in ('left', 'right'):                                                    # :230-240
            for key, src in (('verts3d_MANO_list', v3), ('verts2d_MANO_list', v2)):
                other[key][hand].append(self.converter[hand].GCN_to_vert(src[hand].repeat_interleave(4, dim=1)))
        return (result, {'scale': scale, 'trans2d': trans2d, 'root': root},
                [{'verts3d': v3, 'verts2d': v2}], other)


def load_decoder(opt, encoder_info=None):
    """intaghand_decoder.py:245-278."""
    return decoder(opt, dropout=0.05, num_attn_heads=4)
