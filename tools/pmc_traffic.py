"""Aggregate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py into HBM bytes per launch for the GEMM
family (profiles/rNN_pmc_traffic.json).  Units and corrections per /opt/skills/guides/MI355X_MICROARCH.md (HBM):
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports exactly half the bytes of wide coalesced reads, so it
is doubled; WRITE_SIZE is taken as is.

    python tools/pmc_traffic.py <fetch_dir> <write_dir> <steps_in_trace> <out.json> [tag]

Besides the GEMM family the file carries `hbm_family`: the HBM-bound kernels (PointNet++ data movement, BatchNorm, gathers,
FPS when traced) with counter bytes per launch and, from the dispatch timestamps of the same (serialised) PMC run, the
achieved counter-GB/s -- the north_star's "rocprof counters evidence achieved HBM GB/s on FPS / ball-query".
`gemm_hip_sha256` / `hbm_src_sha256` tie the numbers to the kernel sources they were measured with (bench.py withholds
them when the sources have changed since).
"""
import collections
import csv
import glob
import json
import re
import sys
import hashlib
import os


def tile_name(k):
    """'void igemm_nt<128, 128, 2, 2, true, false>(IGemm)' -> 'igemm_nt<128,128>' (tile shape; the B-layout / fast-path flags merged)."""
    if "igemm_halo3x3" in k:                                   # the LDS-halo form of the 128x128 tile (same tile class as in bench.py)
        return "igemm_nt<128,128>"
    m = re.search(r"(igemm_nt|wgemm_tn_dma|wgemm_tn)<([^>]*)>", k)
    if not m:
        return None
    args = [a.strip() for a in m.group(2).split(",")]
    return "%s<%s>" % (m.group(1), ",".join(args[:2])) if m.group(1) != "wgemm_tn_dma" else "wgemm_tn_dma<128,128>"


def load(d, counter, by_tile=False):
    f = (glob.glob(d + "/*counter_collection.csv") + glob.glob(d + "/*/*counter_collection.csv"))[0]
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"]
        if by_tile:
            fam = tile_name(k)
        else:
            fam = ("igemm_nt" if ("igemm_nt" in k or "igemm_halo3x3" in k or "small_k_gemm" in k or "igemm_dma" in k) else "wgemm_tn" if "wgemm_tn" in k else
                   "reduce_slabs" if ("reduce_slabs" in k or "splitk_finish" in k) else None)
        if fam is None:
            continue
        agg[fam][0] += 1
        agg[fam][1] += float(r["Counter_Value"])
    return agg


HBM_KERNELS = ("knn_ball_group_kernel", "gather_sub_fwd_kernel", "gather_sub_bwd_kernel", "gather_sub_bwd_du_kernel", "gather_sub_bwd_dv_kernel", "invert_index_kernel", "bn_relu_maxk_fwd_kernel", "bn_maxk_bwd_partial_kernel",
               "bn_maxk_bwd_apply_kernel", "gather_rows_kernel", "scatter_rows_kernel", "bn_partial_v4_kernel", "affine_apply_v4_kernel",
               "bn_bwd_partial_v4_kernel", "bn_bwd_apply_v4_kernel", "l2norm_fwd_kernel", "l2norm_bwd_kernel", "l2norm_cat_fwd_kernel", "l2norm_cat_bwd_kernel", "up2_fwd_v4_kernel", "up2_bwd_v4_kernel",
               "adam_kernel", "fps_kernel", "maxk_fwd_kernel", "maxk_bwd_kernel", "group_bwd_kernel")
POINTNET = HBM_KERNELS[:11]


def hbm_family(fdir, wdir, steps):
    def per_kernel(d, counter):
        f = (glob.glob(d + "/*counter_collection.csv") + glob.glob(d + "/*/*counter_collection.csv"))[0]
        agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            m = re.match(r"(?:void )?(\w+)", r["Kernel_Name"])
            k = m.group(1) if m else r["Kernel_Name"]
            if k not in HBM_KERNELS:
                continue
            a = agg[k]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
            if "Start_Timestamp" in r and "End_Timestamp" in r:
                a[2] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
        return agg
    fe, wr = per_kernel(fdir, "FETCH_SIZE"), per_kernel(wdir, "WRITE_SIZE")
    out = {"kernels": {}}
    pn = 0.0
    for k in sorted(set(fe) | set(wr)):
        n = fe[k][0] or wr[k][0]
        rb, wb = fe[k][1] * 1024 * 2, wr[k][1] * 1024
        secs = (fe[k][2] + wr[k][2]) / 2.0 if fe[k][2] and wr[k][2] else (fe[k][2] or wr[k][2])
        out["kernels"][k] = {"launches_per_step": n / steps, "read_MB_per_launch": rb / max(n, 1) / 1e6, "write_MB_per_launch": wb / max(n, 1) / 1e6,
                             "bytes_per_launch": (rb + wb) / max(n, 1), "avg_us": secs / max(n, 1) * 1e6,
                             "counter_GBs": (rb + wb) / secs / 1e9 if secs > 0 else None}
        if k in POINTNET:
            pn += rb + wb
    out["pointnet_bytes_per_step"] = pn / steps
    return out


def bf16_symbols(fdir, wdir, steps, out, batch=64):
    """`--bf16 <fetch_dir> <write_dir> <steps> <out.json> [batch]`: merge the per-symbol HBM bytes of a bf16 run (bench.py --dtype bf16
    --batch 64, or 32) into an existing profile as `bf16_symbols` (`bf16_symbols_B32` for the B=32 run), tied to the bf16 kernel sources by
    `bf16_src_sha256`."""
    key = "bf16_symbols" if batch == 64 else "bf16_symbols_B%d" % batch
    def by_symbol(d, counter):
        f = (glob.glob(d + "/*counter_collection.csv") + glob.glob(d + "/*/*counter_collection.csv"))[0]
        agg = collections.defaultdict(lambda: [0, 0.0])
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter or not re.search(r"gemm|reduce_slabs|splitk", r["Kernel_Name"]):
                continue
            k = re.sub(r"^void |\(.*$", "", r["Kernel_Name"])
            agg[k][0] += 1
            agg[k][1] += float(r["Counter_Value"])
        return agg
    fs, ws = by_symbol(fdir, "FETCH_SIZE"), by_symbol(wdir, "WRITE_SIZE")
    res = json.load(open(out))
    res[key] = {}
    for k in sorted(set(fs) | set(ws)):
        n = fs[k][0] or ws[k][0]
        rb, wb = fs[k][1] * 1024 * 2, ws[k][1] * 1024
        res[key][k] = {"launches_per_step": n / steps, "read_MB_per_launch": rb / max(n, 1) / 1e6, "write_MB_per_launch": wb / max(n, 1) / 1e6,
                                  "bytes_per_launch": (rb + wb) / max(n, 1)}
    res["bf16_run" if batch == 64 else "bf16_run_B%d" % batch] = "bench.py --dtype bf16 --batch %d (separate --pmc FETCH_SIZE / WRITE_SIZE passes, side streams off)" % batch
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = hashlib.sha256()
    for f in ("gemm_bf16.hip", "gemm_dma.hip"):
        h.update(open(os.path.join(root, "pdfnet_amd", "csrc", f), "rb").read())
    res["bf16_src_sha256"] = h.hexdigest()
    json.dump(res, open(out, "w"), indent=1)
    top = sorted(res["bf16_symbols"].items(), key=lambda kv: -kv[1]["bytes_per_launch"] * kv[1]["launches_per_step"])[:8]
    for k, v in top:
        print("bf16 %-60s %5.1f launches/step  read %8.1f MB  write %8.1f MB per launch" % (k[:60], v["launches_per_step"], v["read_MB_per_launch"], v["write_MB_per_launch"]))


def main():
    if sys.argv[1] == "--bf16":
        return bf16_symbols(sys.argv[2], sys.argv[3], int(sys.argv[4]), sys.argv[5], int(sys.argv[6]) if len(sys.argv) > 6 else 64)
    fdir, wdir, steps, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    fe, wr = load(fdir, "FETCH_SIZE"), load(wdir, "WRITE_SIZE")
    res = {"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE -- python3 bench.py (separate passes)",
           "corrections": "KiB -> bytes; FETCH_SIZE x2 (gfx950 wide-read correction, MI355X_MICROARCH.md HBM section)",
           "steps_in_trace": steps, "families": {}}
    tot_b, tot_l = 0.0, 0
    for fam in sorted(set(fe) | set(wr)):
        n = fe[fam][0] or wr[fam][0]
        rb = fe[fam][1] * 1024 * 2
        wb = wr[fam][1] * 1024
        res["families"][fam] = {"launches_per_step": n / steps, "read_GB_per_step": rb / steps / 1e9, "write_GB_per_step": wb / steps / 1e9,
                                "bytes_per_launch": (rb + wb) / max(n, 1)}
        if fam != "reduce_slabs":
            tot_b += rb + wb
            tot_l += n
    res["gemm_family"] = {"hbm_GB_per_step": tot_b / steps / 1e9, "bytes_per_launch": tot_b / max(tot_l, 1)}
    fk, wk = load(fdir, "FETCH_SIZE", True), load(wdir, "WRITE_SIZE", True)
    res["kernels"] = {}
    for k in sorted(set(fk) | set(wk)):
        n = fk[k][0] or wk[k][0]
        b = fk[k][1] * 1024 * 2 + wk[k][1] * 1024
        res["kernels"][k] = {"launches_per_step": n / steps, "hbm_GB_per_step": b / steps / 1e9, "bytes_per_launch": b / max(n, 1)}
    # per rocprofv3 SYMBOL (as bench.py's roofline.kernel names it: 'void ' and the argument list stripped): the bench line looks its
    # dominant symbol up here for `roofline.traffic`
    def by_symbol(d, counter):
        f = (glob.glob(d + "/*counter_collection.csv") + glob.glob(d + "/*/*counter_collection.csv"))[0]
        agg = collections.defaultdict(lambda: [0, 0.0])
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter or not re.search(r"gemm|reduce_slabs|splitk|stem7x7|tiny_conv|narrow_wgrad", r["Kernel_Name"]):
                continue
            k = re.sub(r"^void |\(.*$", "", r["Kernel_Name"])
            agg[k][0] += 1
            agg[k][1] += float(r["Counter_Value"])
        return agg
    fs, ws = by_symbol(fdir, "FETCH_SIZE"), by_symbol(wdir, "WRITE_SIZE")
    res["symbols"] = {}
    for k in sorted(set(fs) | set(ws)):
        n = fs[k][0] or ws[k][0]
        rb, wb = fs[k][1] * 1024 * 2, ws[k][1] * 1024
        res["symbols"][k] = {"launches_per_step": n / steps, "read_MB_per_launch": rb / max(n, 1) / 1e6, "write_MB_per_launch": wb / max(n, 1) / 1e6,
                             "bytes_per_launch": (rb + wb) / max(n, 1)}
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res["tag"] = sys.argv[5] if len(sys.argv) > 5 else ""
    h = hashlib.sha256()                                       # (round 6: gemm.hip + gemm_x3.hip, the two files the fp32-mode GEMM symbols come from)
    for f in ("gemm.hip", "gemm_x3.hip"):
        h.update(open(os.path.join(root, "pdfnet_amd", "csrc", f), "rb").read())
    res["gemm_hip_sha256"] = h.hexdigest()
    h = hashlib.sha256()
    for f in ("pointops.hip", "norm.hip"):
        h.update(open(os.path.join(root, "pdfnet_amd", "csrc", f), "rb").read())
    res["hbm_src_sha256"] = h.hexdigest()
    res["hbm_family"] = hbm_family(fdir, wdir, steps)
    # per-dispatch reads of the 3x3 LDS-halo kernels (the layer VERDICT r01 singled out: `feat` forward fetched 5.75 GB with the
    # per-tap gather) and the heaviest dispatches overall
    f0 = (glob.glob(fdir + "/*counter_collection.csv") + glob.glob(fdir + "/*/*counter_collection.csv"))[0]
    rows0 = [r for r in csv.DictReader(open(f0)) if r["Counter_Name"] == "FETCH_SIZE"]
    halo = collections.defaultdict(list)
    for r in rows0:
        if "igemm_halo3x3" in r["Kernel_Name"]:
            halo[re.sub(r"^void |\(.*$", "", r["Kernel_Name"])].append(round(float(r["Counter_Value"]) * 2 * 1024 / 1e6, 1))
    res["halo3x3_read_MB_per_dispatch"] = {k: sorted(v, reverse=True)[:3 * 8] for k, v in halo.items()}
    rows0.sort(key=lambda r: -float(r["Counter_Value"]))
    res["largest_dispatches_read_MB"] = [[re.sub(r"^void |\(.*$", "", r["Kernel_Name"])[:60], r.get("Grid_Size", "?"),
                                          round(float(r["Counter_Value"]) * 2 * 1024 / 1e6, 1)] for r in rows0[:24]]
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res["gemm_family"]), json.dumps(res["families"]))
    for k, v in res["hbm_family"]["kernels"].items():
        print("%-34s %6.1f launches/step %9.2f MB/launch %8.1f us  %7.1f GB/s (counters)" % (k, v["launches_per_step"], v["bytes_per_launch"] / 1e6, v["avg_us"], v["counter_GBs"]))
    # the heaviest single dispatches (reads), for tile-order / reuse work
    f = (glob.glob(fdir + "/*counter_collection.csv") + glob.glob(fdir + "/*/*counter_collection.csv"))[0]
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == "FETCH_SIZE"]
    rows.sort(key=lambda r: -float(r["Counter_Value"]))
    for r in rows[:16]:
        print("%-64s grid %-8s %8.1f MB read" % (r["Kernel_Name"][:64], r.get("Grid_Size", "?"), float(r["Counter_Value"]) * 2 * 1024 / 1e6))


if __name__ == "__main__":
    main()
