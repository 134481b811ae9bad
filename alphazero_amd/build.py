"""Build libazamd.so for gfx950 with hipcc (cross-compiles without a GPU)."""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = ["az_engine.hip", "az_net.hip", "az_net_mfma.hip", "az_net_layers.hip", "az_replay.hip", "az_hostrng.cpp"]
HDR = ["az_board.cuh", "az_net.h", "az_mfma_gemm.cuh", "az_lane.cuh", "../../include/azamd.h"]
# -ffp-contract=off: the search kernels must round every float64 product and sum separately.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall", "-Wno-unused-function"]


def build(force: bool = False, verbose: bool = False) -> str:
    out = os.path.join(HERE, "libazamd.so")
    csrc = os.path.join(HERE, "csrc")
    deps = [os.path.join(csrc, s) for s in SRC + HDR]
    if not force and os.path.exists(out) and all(os.path.getmtime(out) >= os.path.getmtime(d) for d in deps):
        return out
    objs = []
    procs = []
    for s in SRC:
        o = os.path.join(csrc, os.path.splitext(s)[0] + ".o")
        objs.append(o)
        src = os.path.join(csrc, s)
        if not force and os.path.exists(o) and all(os.path.getmtime(o) >= os.path.getmtime(d) for d in [src] + deps[len(SRC):]):
            continue
        cmd = ["hipcc", *FLAGS, "-c", src, "-o", o] if s.endswith(".hip") else \
              ["hipcc", "-O2", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall", "-c", src, "-o", o]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((s, subprocess.Popen(cmd)))
    for s, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {s}")
    cmd = ["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-pthread", "-o", out, *objs]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
