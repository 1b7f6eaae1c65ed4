"""Build-container tool: convert the reference's graph constants (pickles with scipy/torch objects)
into one plain .npz data asset the module needs at construction (they are not in checkpoints:
`graph_L` is a non-persistent buffer, reference lib/models/networks/model_attn/gcn.py:79-86).

Source files (read-only): /root/reference/lib/models/networks/gcn_core/{graph_left,graph_right,
upsample,v_color}.pkl, loaded by reference intaghand_decoder.py:245-258.
Output: pdfnet_amd/data/gcn_core.npz  (Laplacians as CSR fp32 for V=63/126/252 per hand --
the three levels the decoder uses after reversing the list, intaghand_decoder.py:99-106,125-126 --
plus graph_perm, graph_perm_reverse, upsample 778x252, dense_coor 778x3, mesh_faces).
"""
import os
import pickle
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from oracle.ref_harness import install_stubs, REF_ROOT  # noqa: E402


def main():
    install_stubs()
    d = os.path.join(REF_ROOT, "lib/models/networks/gcn_core")
    out = {}
    for hand in ("left", "right"):
        g = pickle.load(open(os.path.join(d, "graph_%s.pkl" % hand), "rb"))
        for L in g["coarsen_graphs_L"]:
            V = L.shape[0]
            if V not in (63, 126, 252):
                continue
            # same cast as gcn.py:17-31 (coo -> float32 -> dense); CSR keeps the same values
            coo = L.tocoo()
            dense = np.zeros((V, V), np.float32)
            np.add.at(dense, (coo.row, coo.col), coo.data.astype(np.float32))
            rows, cols = np.nonzero(dense)
            indptr = np.zeros(V + 1, np.int32)
            np.add.at(indptr, rows + 1, 1)
            indptr = np.cumsum(indptr).astype(np.int32)
            out["L_%s_%d_data" % (hand, V)] = dense[rows, cols].astype(np.float32)
            out["L_%s_%d_indices" % (hand, V)] = cols.astype(np.int32)
            out["L_%s_%d_indptr" % (hand, V)] = indptr
        out["graph_perm_%s" % hand] = np.asarray(g["graph_perm"], np.int64)
        out["graph_perm_reverse_%s" % hand] = np.asarray(g["graph_perm_reverse"], np.int64)
        out["mesh_faces_%s" % hand] = np.asarray(g["mesh_faces"], np.int32)
    out["upsample"] = np.asarray(pickle.load(open(os.path.join(d, "upsample.pkl"), "rb")), np.float32)
    out["dense_coor"] = np.asarray(pickle.load(open(os.path.join(d, "v_color.pkl"), "rb")), np.float32)
    dst = os.path.join(os.path.dirname(__file__), "..", "pdfnet_amd", "data", "gcn_core.npz")
    np.savez_compressed(dst, **out)
    print("wrote", os.path.abspath(dst), os.path.getsize(dst), "bytes;", len(out), "arrays")


if __name__ == "__main__":
    main()
