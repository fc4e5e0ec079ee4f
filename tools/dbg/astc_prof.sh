#!/bin/bash
# PMC passes over the ASTC kernel (GPU box): bash tools/dbg/astc_prof.sh <tag> <format> <quality>
tag=${1:-a}; fmt=${2:-ASTC_6x6}; q=${3:-3}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/astcprof_$tag
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/tools/bench_formats.py --size 2048 --steps 2 --formats $fmt --qualities $q"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM --kernel-trace --output-format csv -d $OUT/a -o pmc -- $CMD > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SMEM --kernel-trace --output-format csv -d $OUT/b -o pmc -- $CMD > $OUT/b.log 2>&1
python3 - <<PY
import csv, glob, collections
for d in ("a","b"):
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % d, recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
        for row in csv.DictReader(open(f)):
            k = row.get("Kernel_Name","?")[:50]
            if "astc" not in k: continue
            acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
            cnt[(k,row["Counter_Name"])] += 1
        for k, c in acc.items():
            for name, v in c.items():
                print("%-44s %-24s per-dispatch %.6g  (n=%d)" % (k, name, v/cnt[(k,name)], cnt[(k,name)]))
PY
