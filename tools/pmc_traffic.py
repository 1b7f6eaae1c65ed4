"""Aggregate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py into HBM bytes per launch for the GEMM
family (profiles/rNN_pmc_traffic.json).  Units and corrections per /opt/skills/guides/MI355X_MICROARCH.md (HBM):
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports exactly half the bytes of wide coalesced reads, so it
is doubled; WRITE_SIZE is taken as is.

    python tools/pmc_traffic.py <fetch_dir> <write_dir> <steps_in_trace> <out.json>
"""
import collections
import csv
import glob
import json
import re
import sys


def tile_name(k):
    """'void igemm_nt<128, 128, 2, 2, true, false>(IGemm)' -> 'igemm_nt<128,128>' (tile shape; the B-layout / fast-path flags merged)."""
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
            fam = "igemm_nt" if "igemm_nt" in k else "wgemm_tn" if "wgemm_tn" in k else "reduce_slabs" if "reduce_slabs" in k else None
        if fam is None:
            continue
        agg[fam][0] += 1
        agg[fam][1] += float(r["Counter_Value"])
    return agg


def main():
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
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res["gemm_family"]), json.dumps(res["families"]))
    # the heaviest single dispatches (reads), for tile-order / reuse work
    f = (glob.glob(fdir + "/*counter_collection.csv") + glob.glob(fdir + "/*/*counter_collection.csv"))[0]
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == "FETCH_SIZE"]
    rows.sort(key=lambda r: -float(r["Counter_Value"]))
    for r in rows[:16]:
        print("%-64s grid %-8s %8.1f MB read" % (r["Kernel_Name"][:64], r.get("Grid_Size", "?"), float(r["Counter_Value"]) * 2 * 1024 / 1e6))


if __name__ == "__main__":
    main()
