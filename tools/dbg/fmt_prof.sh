#!/bin/bash
# PMC pass over one format's kernel (GPU box): bash tools/dbg/fmt_prof.sh <tag> <format> <quality> <kernel substring>
tag=${1:-a}; fmt=${2:-ETC2_R8G8B8}; q=${3:-2}; pat=${4:-etc}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/fmtprof_$tag
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/tools/bench_formats.py --size 2048 --steps 2 --formats $fmt --qualities $q"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM --kernel-trace --output-format csv -d $OUT/a -o pmc -- $CMD > $OUT/a.log 2>&1
grep format $OUT/a.log
python3 - <<PY
import csv, glob, collections
for f in glob.glob("$OUT/a/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name","?")[:60]
        if "$pat" not in k: continue
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"]); cnt[(k,row["Counter_Name"])] += 1
    for k, c in acc.items():
        for name, v in c.items():
            print("%-50s %-22s per-dispatch %.6g" % (k, name, v/cnt[(k,name)]))
PY
