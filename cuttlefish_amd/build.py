"""Build the gfx950 shared library (C-ABI of include/cuttlefish_hip.h) in-tree with hipcc.

hipcc cross-compiles --offload-arch=gfx950 without a GPU; the resulting
cuttlefish_amd/libcuttlefish_hip.so travels to the GPU box with the snapshot.
"""
from __future__ import annotations

import os
import shutil
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "libcuttlefish_hip.so")

# -ffp-contract=off: the kernels spell fused ops as fmaf() so that every float
# operation matches the CPU oracle bit for bit.
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC",
               "-shared", "-Wall", "-Wno-unused-function"]


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _deps():
    d = [os.path.join(CSRC, f) for f in os.listdir(CSRC)]
    d.append(os.path.join(os.path.dirname(PKG), "include", "cuttlefish_hip.h"))
    return d


def _fingerprint() -> str:
    """SHA-256 over the build flags and the CONTENT of every dependency: modification times are
    not trusted (git checkout / stash restore files with arbitrary times, and a stale library on
    the GPU box silently measures old kernels)."""
    import hashlib
    h = hashlib.sha256(" ".join(HIPCC_FLAGS).encode())
    for p in sorted(_deps()):
        if os.path.isfile(p):
            h.update(os.path.basename(p).encode())
            with open(p, "rb") as f:
                h.update(f.read())
    return h.hexdigest()


STAMP = LIB + ".srchash"


def is_stale() -> bool:
    if not os.path.exists(LIB) or not os.path.exists(STAMP):
        return True
    with open(STAMP) as f:
        return f.read().strip() != _fingerprint()


def hipcc_path():
    return shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not is_stale():
        return LIB
    cmd = [hipcc_path()] + HIPCC_FLAGS + ["-o", LIB + ".tmp"] + sources()
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    os.replace(LIB + ".tmp", LIB)
    with open(STAMP, "w") as f:
        f.write(_fingerprint() + "\n")
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
