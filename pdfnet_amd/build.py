"""Build libpdfnet_hip.so (gfx950) in-tree with hipcc.  `python -m pdfnet_amd.build [--force]`."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libpdfnet_hip.so")
SOURCES = ["gemm.hip", "gemm_dma.hip", "gemm_x3.hip", "winograd.hip", "gemm_bf16.hip", "pointops.hip", "norm.hip", "elementwise.hip", "graph.hip", "meshdec.hip", "meshdec_bf16.hip", "meshdec_x3.hip", "mano.hip", "frontend.hip", "loss.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-fvisibility=hidden", "-Wno-unused-result"]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    hdrs = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "gemm_common.h")]
    objs, jobs = [], []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(CSRC, s.replace(".hip", ".o"))
        objs.append(obj)
        deps = [src] + hdrs + ([os.path.join(CSRC, "meshdec.hip")] if s in ("meshdec_bf16.hip", "meshdec_x3.hip") else [])     # (it #includes meshdec.hip)
        if force or _stale(obj, deps):
            jobs.append([hipcc] + FLAGS + ["-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    if force or jobs or _stale(LIB, objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
