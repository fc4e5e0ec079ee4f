#!/usr/bin/env python3
"""profiles/astc_c3_pmc.json from the round's PMC passes of BASELINE config 3 (ASTC 6x6 High, 4096x4096): what
`bench.py --config c3` replays beside its timed numbers (PMC passes cannot run inside a timed benchmark).

    python tools/astc_pmc_to_json.py gpurun_out/r06_astc_pmc_summary.txt gpurun_out/r06_astc_lds_pmc.txt r06
"""
import json
import os
import re
import sys

summ, lds, tag = sys.argv[1], sys.argv[2], sys.argv[3]
K = "cfhip_astc_encode_kernel<0, 12, false>"
vals = {}
for ln in open(summ):
    m = re.match(r"void (cfhip_astc_encode_kernel<[^>]+>)\(cf_kparams\)\s+(\S+)\s+total \S+\s+per-dispatch (\S+)", ln)
    if m and m.group(1) == K:
        vals[m.group(2)] = float(m.group(3))
busy = conflict = wait = None
for ln in open(lds):
    if ln.startswith("void " + K[:40]):
        conflict = float(re.search(r"conflict/active (\S+)", ln).group(1))
        wait = float(re.search(r"wait_any/wave_cycles (\S+)", ln).group(1))
        busy = float(ln.strip().split()[-1])
out = {"source": "profiles/%s_astc_pmc_summary.txt, profiles/%s_astc_lds_pmc.txt" % (tag, tag), "kernel": K,
       "traffic_bytes_per_launch": int((2 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024),      # gfx950: FETCH_SIZE counts 2 KiB units
       "valu_wave_insts_per_launch": int(vals["SQ_INSTS_VALU"]),
       "valu_busy": busy, "lds_bank_conflict_share": conflict, "wait_any_share": wait,
       "valu_busy_note": "SQ_ACTIVE_INST_VALU x 4 cycles / (1024 SIMDs x busy cycles): the share of its SIMD cycles in which the kernel has a "
                         "vector instruction in flight"}
path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "astc_c3_pmc.json")
json.dump(out, open(path, "w"), indent=1)
print(out)
