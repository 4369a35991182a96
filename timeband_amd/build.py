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


EXT = os.path.join(HERE, "_tbext.so")
EXT_SRC = os.path.join(CSRC, "ext.cpp")


def ext_needs_build():
    if not os.path.exists(EXT):
        return True
    t = os.path.getmtime(EXT)
    hdr = os.path.join(os.path.dirname(HERE), "include", "timeband_hip.h")
    return any(os.path.getmtime(d) > t for d in (EXT_SRC, hdr))


def build_ext(force=False, verbose=True):
    """The thin PyTorch-ROCm C++ extension (csrc/ext.cpp: torch::autograd::Functions over the C-ABI) -> timeband_amd/_tbext.so.
    Host C++ only (g++): it contains no kernels and links against libtimeband_hip.so next to it."""
    if not force and not ext_needs_build():
        return EXT
    import sysconfig
    import torch
    from torch.utils import cpp_extension as ce
    tlib = os.path.join(os.path.dirname(torch.__file__), "lib")
    inc = [f"-I{p}" for p in ce.include_paths()] + [f"-I{sysconfig.get_paths()['include']}", "-I/opt/rocm/include",
                                                    f"-I{os.path.join(os.path.dirname(HERE), 'include')}"]
    cxx = shutil.which("g++") or "g++"
    cmd = [cxx, "-O2", "-fPIC", "-shared", "-std=c++17", "-Wall", "-Wno-unused-function", "-Wno-sign-compare",
           "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1", "-DTORCH_EXTENSION_NAME=_tbext", "-DTORCH_API_INCLUDE_EXTENSION_H",
           f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}", *inc, EXT_SRC, "-o", EXT,
           f"-L{tlib}", "-lc10", "-lc10_hip", "-ltorch_cpu", "-ltorch_hip", "-ltorch", "-ltorch_python",
           f"-L{HERE}", "-l:libtimeband_hip.so", "-Wl,-rpath,$ORIGIN", f"-Wl,-rpath,{tlib}"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return EXT


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
    if "--stamps" not in sys.argv:
        print(build_ext(force="--force" in sys.argv))
