#!/bin/bash
# LDS bank-conflict share and wait share of the ASTC kernel at BASELINE config 3 (GPU box): one PMC pass
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/prof_astc_lds
mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT -o pmc -- \
  python $R/tools/bench_formats.py --size 4096 --steps 2 --formats ASTC_6x6 --qualities 3 > $OUT/log.txt 2>&1
python3 - <<PY
import csv, glob, collections
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for row in csv.DictReader(open(f)):
        acc[row["Kernel_Name"][:48]][row["Counter_Name"]] += float(row["Counter_Value"])
    for k, c in acc.items():
        if "LDS" in "".join(c):
            print(k, "conflict/active %.3f" % (c["SQ_LDS_BANK_CONFLICT"]/max(c["SQ_LDS_IDX_ACTIVE"],1)), "wait_any/wave_cycles %.3f" % (c["SQ_WAIT_ANY"]/max(c["SQ_WAVE_CYCLES"],1)),
                  "valu busy (ACTIVE_INST_VALU x 4 / (BUSY_CYCLES/32 x 1024 SIMD)) %.3f" % (c["SQ_ACTIVE_INST_VALU"]*4/max(c["SQ_BUSY_CYCLES"]/32*1024,1)))
PY
