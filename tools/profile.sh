#!/bin/bash
# Run on the GPU box:  bash tools/profile.sh <tag> [bench args...]
# Collects (1) rocprofv3 kernel-trace stats and (2..4) separate PMC passes for bench.py and
# writes the summaries under gpurun_out/prof_<tag>/ (copy what matters into profiles/).
tag=${1:-r1}; shift
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/prof_$tag
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH=${PROFILE_CMD:-"python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-end-to-end --no-second-tile --no-tolerance $@"}   # PROFILE_CMD: another driver (run from /tmp)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o trace -- $BENCH > $OUT/stats.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM --kernel-trace --output-format csv -d $OUT/pmc_sq -o pmc -- $BENCH > $OUT/pmc_sq.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o pmc -- $BENCH > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o pmc -- $BENCH > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $OUT/pmc_sq2 -o pmc -- $BENCH > $OUT/pmc_sq2.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT64 --kernel-trace --output-format csv -d $OUT/pmc_class -o pmc -- $BENCH > $OUT/pmc_class.log 2>&1
find $OUT -name "*.csv" | head -50
for f in $(find $OUT/stats -name "*kernel_stats.csv"); do echo "--- $f"; head -8 $f; done
python3 - <<PY
import csv, glob, collections
for d in ("pmc_sq","pmc_fetch","pmc_write","pmc_sq2","pmc_class"):
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % d, recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
        for row in csv.DictReader(open(f)):
            k = row.get("Kernel_Name","?")[:60]
            acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
            cnt[(k,row["Counter_Name"])] += 1
        print("---", d)
        for k, c in acc.items():
            for name, v in c.items():
                print("%-60s %-24s total %.6g  per-dispatch %.6g  (n=%d)" % (k, name, v, v/cnt[(k,name)], cnt[(k,name)]))
PY
