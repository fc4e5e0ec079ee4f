#!/usr/bin/env python3
"""Fold the PMC passes of tools/profile.sh into profiles/bc7_pmc.json, the file bench.py reads
for `roofline.traffic` and the VALU-issue view (matched by the kernel's code hash, so a stale
file is never quoted for other code).

    python tools/pmc_to_json.py gpurun_out/prof_<tag> --tag r02a [--kernel bc7_encode_kernel<0, true, false>]

Run where the library the passes profiled is the built one (the hash comes from it).
"""
import argparse
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa_mix  # noqa: E402


def per_dispatch(prof_dir, kernel_substr):
    acc, cnt = collections.defaultdict(float), collections.Counter()
    for f in glob.glob(os.path.join(prof_dir, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if kernel_substr not in row.get("Kernel_Name", ""):
                continue
            key = (os.path.relpath(f, prof_dir).split(os.sep)[0], row["Counter_Name"])
            acc[key] += float(row["Counter_Value"])
            cnt[key] += 1
    out = {}
    for (d, name), v in acc.items():
        out.setdefault(name, v/cnt[(d, name)])     # first pass that has the counter
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("prof_dir")
    ap.add_argument("--tag", required=True)
    ap.add_argument("--kernel", default="cfhip_bc7_encode_kernel<0, true, false>")
    ap.add_argument("--mangled", default="cfhip_bc7_encode_kernelILi0ELb1ELb0E")
    ap.add_argument("--quality", type=int, default=2, help="Texture::Quality the passes ran at")
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "bc7_pmc.json"))
    a = ap.parse_args()
    c = per_dispatch(a.prof_dir, a.kernel)
    if "SQ_INSTS_VALU" not in c:
        raise SystemExit("no SQ_INSTS_VALU rows for %r under %s" % (a.kernel, a.prof_dir))
    mix = isa_mix.kernel_mix(os.path.join(ROOT, "cuttlefish_amd", "libcuttlefish_hip.so"), a.mangled)
    insts = c["SQ_INSTS_VALU"]
    out = {
        "kernel": a.kernel, "build": a.tag, "quality": a.quality, "code_sha256": mix["code_sha256"],
        "source": "profiles/%s_bc7_pmc_summary.txt (rocprofv3 --pmc passes of tools/profile.sh, one counter group per run)" % a.tag,
        "valu_wave_insts_per_launch": insts,
        "salu_wave_insts_per_launch": c.get("SQ_INSTS_SALU"),
        "lds_wave_insts_per_launch": c.get("SQ_INSTS_LDS"),
    }
    fp = [c.get(k) for k in ("SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_ADD_F32")]
    if all(v is not None for v in fp):
        # dynamic fp32 fma/mul/add counts at 2 cycles, the remainder at the static non-fp32 price
        fast = sum(fp)
        out["valu_class_counts_per_launch"] = {k: c[k] for k in c if k.startswith("SQ_INSTS_VALU_")}
        out["valu_issue_cycles_per_launch"] = 2.0*fast + (insts - fast)*mix["valu_cycles_per_non_fp32_inst"]
        out["class_source"] = ("dynamic SQ_INSTS_VALU_FMA_F32/_MUL_F32/_ADD_F32 at 2 cycles; the other %.3g wave "
                               "instructions at the static non-fp32 price of %.3f cycles (tools/isa_mix.py)"
                               % (insts - fast, mix["valu_cycles_per_non_fp32_inst"]))
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        out["fetch_size_kb_per_launch"] = c["FETCH_SIZE"]
        out["write_size_kb_per_launch"] = c["WRITE_SIZE"]
        out["fetch_correction"] = 2.0
        out["traffic_bytes_per_launch"] = int(2.0*c["FETCH_SIZE"]*1024 + c["WRITE_SIZE"]*1024)
        out["traffic_note"] = ("gfx950 FETCH_SIZE tallies 128-B requests at 64 B (MI355X_MICROARCH.md, HBM section): "
                               "doubled; WRITE_SIZE as is (equals the payload)")
    json.dump(out, open(a.out, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
