#!/usr/bin/env python3
"""Instruction-class histogram of a kernel's ISA and the mix-weighted VALU issue cost.

The block encoders are VALU-issue-bound, and gfx950 issues wave64 VALU instructions in two
classes (tools/ubench/valu_rate.hip, profiles/r01_valu_rate.txt): fp32 fma / mul / add, plain
v_add / v_sub / v_and / v_or / v_xor / v_mov / v_ashrrev in 2 cycles per SIMD, everything else
this path uses (dot4, shifts, shift-adds, bit-field ops, multiplies, min/max3, conversions,
cndmask, DPP moves, 64-bit mads) in 4.  A roofline fraction quoted against ONE class rate is not
a ceiling; this tool prices the kernel's own instruction mix:

    cycles per VALU instruction = sum over opcodes (static share x class cycles)

taken from the device code object INSIDE the built library (llvm-objcopy + the clang offload
bundle format + llvm-objdump), so it always describes the binary that runs.  The static mix
stands in for the dynamic one (the hot loops dominate both); bench.py multiplies it by the
measured SQ_INSTS_VALU of the committed PMC pass.

    python tools/isa_mix.py [--lib cuttlefish_amd/libcuttlefish_hip.so] [--kernel bc7_encode] [--json out]
"""
import argparse
import collections
import hashlib
import json
import os
import re
import struct
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
# 2-cycle class (measured): everything else VALU is priced at 4
FAST = {"v_add_f32", "v_sub_f32", "v_mul_f32", "v_fma_f32", "v_fmac_f32", "v_mac_f32", "v_add_u32",
        "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_mov_b32",
        "v_ashrrev_i32", "v_add_co_u32", "v_sub_co_u32", "v_not_b32", "v_max_f32", "v_min_f32"}


def device_objects(lib):
    """-> list of gfx950 code objects (bytes) embedded in the library's .hip_fatbin section"""
    with tempfile.TemporaryDirectory() as td:
        fat = os.path.join(td, "fat.bin")
        subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "--dump-section",
                               ".hip_fatbin=" + fat, lib, os.path.join(td, "copy.so")])
        data = open(fat, "rb").read()
    out = []
    pos = data.find(MAGIC)
    while pos >= 0:
        n = struct.unpack_from("<Q", data, pos + len(MAGIC))[0]
        p = pos + len(MAGIC) + 8
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", data, p)
            triple = data[p + 24:p + 24 + tlen].decode()
            p += 24 + tlen
            if "gfx950" in triple and size:
                out.append(data[pos + off:pos + off + size])
        pos = data.find(MAGIC, pos + len(MAGIC))
    return out


def kernel_mix(lib, kernel_substr):
    best = None
    for co in device_objects(lib):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(co)
            f.flush()
            txt = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", f.name],
                                 capture_output=True, text=True).stdout
        cur, body = None, collections.defaultdict(list)
        for line in txt.splitlines():
            m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
            if m:
                cur = m.group(1)
                continue
            if cur and line.startswith("\t"):
                op = line.split()[0]
                body[cur].append(op)
        for name, ops in body.items():
            if kernel_substr in name and (best is None or len(ops) > len(best[1])):
                best = (name, ops)
    if best is None:
        raise SystemExit("no kernel matching %r in %s" % (kernel_substr, lib))
    name, ops = best
    hist = collections.Counter(ops)
    valu = {k: v for k, v in hist.items() if k.startswith("v_") and not k.startswith("v_readlane")
            and not k.startswith("v_readfirstlane") and not k.startswith("v_writelane")}
    total = sum(valu.values())
    fast = sum(v for k, v in valu.items() if re.sub(r"_(e32|e64|dpp|sdwa|e64_dpp)$", "", k) in FAST)
    cpi = (2.0*fast + 4.0*(total - fast))/total
    # the same price without the fp32 fma/mul/add/min/max opcodes: the PMC class counters count those
    # dynamically (SQ_INSTS_VALU_FMA_F32 / _MUL_F32 / _ADD_F32), the static mix prices only the rest
    fp = sum(v for k, v in valu.items() if re.sub(r"_(e32|e64|dpp|sdwa|e64_dpp)$", "", k) in FAST and k.split("_")[-2 if k.endswith(("_e32", "_e64", "_dpp", "_sdwa")) else -1] == "f32")
    rest_total, rest_fast = total - fp, fast - fp
    cpi_rest = (2.0*rest_fast + 4.0*(rest_total - rest_fast))/max(rest_total, 1)
    return {
        "kernel": name,
        "code_sha256": hashlib.sha256(" ".join(ops).encode()).hexdigest()[:16],
        "instructions": len(ops), "valu": total, "salu": sum(v for k, v in hist.items() if k.startswith("s_")),
        "lds": sum(v for k, v in hist.items() if k.startswith("ds_")),
        "valu_fast_share": round(fast/total, 4),
        "valu_cycles_per_inst": round(cpi, 4),
        "valu_fp32_fast_share": round(fp/total, 4),
        "valu_cycles_per_non_fp32_inst": round(cpi_rest, 4),
        "top_valu": sorted(valu.items(), key=lambda kv: -kv[1])[:16],
        "classes": "2 cycles: " + ", ".join(sorted(FAST)) + "; 4 cycles: every other VALU opcode "
                   "(profiles/r01_valu_rate.txt)",
    }


def main():
    ap = argparse.ArgumentParser()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ap.add_argument("--lib", default=os.path.join(root, "cuttlefish_amd", "libcuttlefish_hip.so"))
    ap.add_argument("--kernel", default="cfhip_bc7_encode_kernelILi0ELb1ELb0E")
    ap.add_argument("--json", default="")
    a = ap.parse_args()
    r = kernel_mix(a.lib, a.kernel)
    print(json.dumps(r, indent=1))
    if a.json:
        json.dump(r, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
