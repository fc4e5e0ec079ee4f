#!/bin/bash
# LDS / wait counters of the ASTC kernel (GPU box): bash tools/dbg/astc_lds_prof.sh [format] [quality]
fmt=${1:-ASTC_6x6}; q=${2:-3}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/astc_lds
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/tools/bench_formats.py --size 2048 --steps 2 --formats $fmt --qualities $q"
for grp in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" "SQ_INST_CYCLES_SALU SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAIT_ANY"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/$tag -o pmc -- $CMD > $OUT/$tag.log 2>&1
  python3 - <<PY
import csv, glob, collections
for f in glob.glob("$OUT/$tag/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(float); cnt = collections.Counter()
    for row in csv.DictReader(open(f)):
        if "astc" not in row.get("Kernel_Name",""): continue
        if "<0" not in row.get("Kernel_Name","") and "ILi0" not in row.get("Kernel_Name",""): continue
        acc[row["Counter_Name"]] += float(row["Counter_Value"]); cnt[row["Counter_Name"]] += 1
    for name, v in acc.items():
        print("%-26s per-dispatch %.6g" % (name, v/cnt[name]))
PY
done
