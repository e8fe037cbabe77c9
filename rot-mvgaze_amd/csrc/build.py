#!/usr/bin/env python3
"""Build librotmvgaze_hip.so (gfx950) in-tree with hipcc.  Used by __graft_entry__.build()."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = ["api.hip", "conv_igemm.hip", "conv_bf16.hip", "conv_split.hip", "bn.hip", "pool.hip", "fusion.hip", "pair_index.cpp"]
LIB = os.path.join(HERE, "librotmvgaze_hip.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function", "-Wno-unused-value", "-Wno-unused-result",
         "-fno-gpu-rdc"]


def _stale(obj, src):
    if not os.path.exists(obj):
        return True
    deps = [src, os.path.join(HERE, "common.h"), os.path.join(HERE, "conv_shared.h"),
            os.path.join(HERE, "elem.h"), os.path.join(HERE, "bf16_tile.h"), os.path.join(HERE, "..", "..", "include", "rotmvgaze.h"),
            os.path.abspath(__file__)]
    return any(os.path.exists(d) and os.path.getmtime(d) > os.path.getmtime(obj) for d in deps)


def build(force=False, verbose=True):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs, jobs = [], []
    for s in SOURCES:
        src = os.path.join(HERE, s)
        obj = os.path.join(HERE, os.path.splitext(s)[0] + ".o")
        objs.append(obj)
        if force or _stale(obj, src):
            cmd = [hipcc] + FLAGS + (["-x", "hip"] if s.endswith(".hip") else []) + ["-c", src, "-o", obj]
            jobs.append(cmd)

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    if jobs or not os.path.exists(LIB):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
