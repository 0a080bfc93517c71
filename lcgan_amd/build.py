"""Builds lcgan_amd/liblcgan_hip.so for gfx950 with hipcc (in-tree; the .so is git-ignored but travels to the GPU box)."""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "liblcgan_hip.so")
SOURCES = ["conv_igemm.hip", "conv_fp8.hip", "stencil.hip", "small.hip", "views.hip", "prof.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-munsafe-fp-atomics", "-Wno-unused-value"]


STAMP = LIB + ".srchash"      # content hash of the sources the library was built from (travels with the .so; mtimes do not survive a copy)


def source_hash() -> str:
    import hashlib
    h = hashlib.sha256(" ".join(FLAGS).encode())
    for f in sorted(os.listdir(CSRC)):
        if f.endswith((".hip", ".h")):
            with open(os.path.join(CSRC, f), "rb") as fh:
                h.update(f.encode() + b"\0" + fh.read())
    return h.hexdigest()


def _stale() -> bool:
    if not (os.path.exists(LIB) and os.path.exists(STAMP)):
        return True
    with open(STAMP) as f:
        return f.read().strip() != source_hash()


def build(force: bool = False, verbose: bool = True) -> str:
    import fcntl
    with open(LIB + ".lock", "w") as lock:           # ranks of one node may reach this together: one builds, the others wait
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not force and not _stale():
            return LIB
        return _build_locked(verbose)


def _build_locked(verbose: bool) -> str:
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    digest = source_hash()
    objs = []
    for src in SOURCES:
        obj = os.path.join(CSRC, src.replace(".hip", ".o"))
        cmd = [hipcc, *FLAGS, "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        objs.append(obj)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    with open(STAMP, "w") as f:
        f.write(digest)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
