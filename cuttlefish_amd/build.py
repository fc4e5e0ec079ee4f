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


def _object_key(src: str) -> str:
    """Content hash of one translation unit: flags, the source, every header of csrc/ and the ABI header."""
    import hashlib
    h = hashlib.sha256(" ".join(HIPCC_FLAGS).encode())
    deps = [src] + sorted(p for p in _deps() if p.endswith(".h"))
    for p in deps:
        h.update(os.path.basename(p).encode())
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def build(force: bool = False, verbose: bool = False) -> str:
    """One object per .hip file (compiled in parallel, reused while its content hash stands:
    changing one kernel recompiles one file), then one link."""
    if not force and not is_stale():
        return LIB
    from concurrent.futures import ThreadPoolExecutor
    objdir = os.path.join(PKG, "build")
    os.makedirs(objdir, exist_ok=True)
    cflags = [f for f in HIPCC_FLAGS if f != "-shared"]

    def compile_one(src):
        base = os.path.splitext(os.path.basename(src))[0]
        obj = os.path.join(objdir, base + ".o")
        stamp = obj + ".key"
        key = _object_key(src)
        if not force and os.path.exists(obj) and os.path.exists(stamp):
            with open(stamp) as f:
                if f.read().strip() == key:
                    return obj
        cmd = [hipcc_path()] + cflags + ["-c", "-o", obj + ".tmp", src]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        os.replace(obj + ".tmp", obj)
        with open(stamp, "w") as f:
            f.write(key + "\n")
        return obj

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        objs = list(ex.map(compile_one, sources()))
    # the link line comes from the same flag list as the compiles (arch and anything link-relevant stay in step)
    lflags = [f for f in HIPCC_FLAGS if f.startswith(("--offload-arch", "-fgpu-rdc", "-fsanitize")) or f in ("-shared", "-fPIC")]
    cmd = [hipcc_path()] + lflags + ["-o", LIB + ".tmp"] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    os.replace(LIB + ".tmp", LIB)
    with open(STAMP, "w") as f:
        f.write(_fingerprint() + "\n")
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
