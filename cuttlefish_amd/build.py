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
    check_no_vector_spills(LIB + ".tmp")          # a spilling build never becomes the library
    os.replace(LIB + ".tmp", LIB)
    with open(STAMP, "w") as f:
        f.write(_fingerprint() + "\n")
    return LIB


LLVM_BIN = "/opt/rocm/lib/llvm/bin"
BLOCK_KERNELS = ("cfhip_bc7_encode_kernel", "cfhip_bc15_encode_kernel", "cfhip_bc6h_encode_kernel",
                 "cfhip_etc_encode_kernel", "cfhip_astc_encode_kernel")


def kernel_metadata(lib: str = LIB):
    """{kernel name: {scratch, vgpr, agpr, vgpr_spill, sgpr_spill, lds}} read from the gfx950 code objects inside the
    built library (llvm-objdump --offloading + llvm-readelf --notes); None when the ROCm LLVM tools are absent."""
    import re
    import tempfile
    objdump, readelf = os.path.join(LLVM_BIN, "llvm-objdump"), os.path.join(LLVM_BIN, "llvm-readelf")
    if not (os.path.exists(lib) and os.path.exists(objdump) and os.path.exists(readelf)):
        return None
    out = {}
    with tempfile.TemporaryDirectory() as d:
        cp = shutil.copy(lib, d)                       # llvm-objdump --offloading extracts next to its input
        subprocess.run([objdump, "--offloading", cp], cwd=d, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        for f in sorted(os.listdir(d)):
            if "amdgcn" not in f:
                continue
            assert f.endswith("gfx950"), f             # one target, no fat binary
            notes = subprocess.run([readelf, "--notes", os.path.join(d, f)], check=True, capture_output=True, text=True).stdout
            # one YAML list item per kernel ("  - .agpr_count: ..." opens it: the keys are sorted, .name comes later)
            for item in re.split(r"\n\s+- (?=\.agpr_count:|\.args:)", notes):
                name = re.search(r"\.name:\s+(\S+)", item)
                if not name or ".vgpr_count" not in item:
                    continue

                def g(k):
                    m = re.search(r"\.%s:\s+(\d+)" % k, item)
                    return int(m.group(1)) if m else None
                out[name.group(1)] = {"scratch": g("private_segment_fixed_size"), "vgpr": g("vgpr_count"), "agpr": g("agpr_count"),
                                      "vgpr_spill": g("vgpr_spill_count"), "sgpr_spill": g("sgpr_spill_count"),
                                      "lds": g("group_segment_fixed_size")}
    return out


def check_no_vector_spills(lib: str = LIB):
    """A correctness rule, not a performance one (round 5): these kernels read each other's lanes (DPP, ds_bpermute,
    v_readlane), and a vector register spilled inside divergent control flow saves the active lanes only -- a spilling
    ETC2 build returned wrong blocks.  On gfx950 the compiler can also park VGPRs in AGPRs without touching scratch, so
    scratch == 0 alone does not prove it: every block kernel must show vgpr_spill_count 0, agpr_count 0 and no scratch.
    Raises RuntimeError naming the offenders; silently passes where the LLVM tools are missing."""
    meta = kernel_metadata(lib)
    if meta is None:
        return
    bad = {k: v for k, v in meta.items() if any(s in k for s in BLOCK_KERNELS) and
           (v["scratch"] != 0 or (v["vgpr_spill"] or 0) != 0 or (v["agpr"] or 0) != 0)}
    if bad:
        raise RuntimeError("block kernels with spilled vector registers (wrong blocks under divergence): %r" % (bad,))


if __name__ == "__main__":
    print(build(force=True, verbose=True))
