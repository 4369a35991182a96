"""Build the gfx950 shared library (C-ABI, include/timeband_hip.h) in-tree with hipcc.

    python -m timeband_amd.build [--force]

hipcc cross-compiles for gfx950 without a GPU.  The .so lands next to this file
(timeband_amd/libtimeband_hip.so), is git-ignored, and travels to the GPU box with the tree.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libtimeband_hip.so")
SOURCES = ["capi.hip", "gemm.hip", "lstm.hip", "losses.hip", "postproc.hip", "head.hip"]
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function"]


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the HIP library cannot be built")
    return exe


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True, jobs=6, stamps=False):
    """stamps=True: diagnostic build with in-kernel cycle stamps (libtimeband_hip_stamps.so, tools only)."""
    if stamps:
        return _build(os.path.join(HERE, "build_stamps"), os.path.join(HERE, "libtimeband_hip_stamps.so"),
                      FLAGS + ["-DTB_STAMPS"] + os.environ.get("TB_EXTRA_FLAGS", "").split(), verbose, jobs)
    if not force and not needs_build():
        return LIB
    return _build(os.path.join(HERE, "build"), LIB, FLAGS + os.environ.get("TB_EXTRA_FLAGS", "").split(), verbose, jobs)


def _build(objdir, LIB, FLAGS, verbose, jobs):
    hipcc = _hipcc()
    os.makedirs(objdir, exist_ok=True)
    procs = []
    objs = []
    for src in SOURCES:
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        objs.append(obj)
        cmd = [hipcc, *FLAGS, "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        if len(procs) >= jobs:
            _drain(procs)
    _drain(procs)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return LIB


def _drain(procs):
    while procs:
        src, p = procs.pop(0)
        out, _ = p.communicate()
        if out.strip():
            print(out, flush=True)
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}")


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, stamps="--stamps" in sys.argv))
