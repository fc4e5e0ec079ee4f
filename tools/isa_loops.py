#!/usr/bin/env python3
"""Loop table of one kernel in a hipcc -S listing (tools/kres.sh ... -gline-tables-only): every
backward branch = a loop; prints its source line range, instruction counts by class (VALU, SALU,
LDS, scratch / global) and whether it contains an inner loop.  A reading aid for instruction-count
work on the issue-bound encoders.
    tools/isa_loops.py /tmp/astc_encode.s <kernel substring> [min instructions]
"""
import re
import sys

path, sub = sys.argv[1], sys.argv[2]
minsz = int(sys.argv[3]) if len(sys.argv) > 3 else 8
lines = open(path).read().splitlines()
start = next(i for i, l in enumerate(lines) if re.match(r"^[_A-Za-z0-9]+:", l) and sub in l)
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
body = lines[start:end + 1]
label_at, insts = {}, []       # insts: (text, srcline)
cur_src = 0
for l in body:
    s = l.strip()
    m = re.match(r"^(\.LBB[0-9_]+):", s)
    if m:
        label_at[m.group(1)] = len(insts)
        continue
    m = re.match(r"^\.loc\s+\d+\s+(\d+)", s)
    if m:
        cur_src = int(m.group(1))
        continue
    if not s or s.startswith((".", ";", "//")) or s.endswith(":"):
        continue
    insts.append((s.split(";")[0].strip(), cur_src))
loops = []
for i, (t, _) in enumerate(insts):
    m = re.match(r"^s_cbranch_\w+\s+(\.LBB[0-9_]+)", t) or re.match(r"^s_branch\s+(\.LBB[0-9_]+)", t)
    if m and m.group(1) in label_at and label_at[m.group(1)] <= i:
        loops.append((label_at[m.group(1)], i))
loops = sorted(set(loops))
def cls(t):
    op = t.split()[0]
    if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")): return "xl"
    if op.startswith("v_"): return "valu"
    if op.startswith("s_"): return "salu"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("scratch_", "global_", "buffer_", "flat_")): return "mem"
    return "other"
print("%-13s %-11s %5s %5s %5s %4s %4s %4s  %s" % ("insts", "src lines", "total", "valu", "salu", "lds", "mem", "xl", "inner"))
for a, b in loops:
    n = b - a + 1
    if n < minsz:
        continue
    c = {}
    for t, _ in insts[a:b + 1]:
        c[cls(t)] = c.get(cls(t), 0) + 1
    src = [s for _, s in insts[a:b + 1] if s]
    inner = sum(1 for (x, y) in loops if a <= x and y <= b and (x, y) != (a, b))
    print("%6d-%-6d %5d-%-5d %5d %5d %5d %4d %4d %4d  %d" % (a, b, min(src) if src else 0, max(src) if src else 0, n,
          c.get("valu", 0), c.get("salu", 0), c.get("lds", 0), c.get("mem", 0), c.get("xl", 0), inner))
