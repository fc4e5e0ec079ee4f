#!/bin/bash
# Extra PMC passes for the front-end question (instruction fetch / branches / issue mix):
#   bash tools/pmc_extra.sh <tag>
tag=${1:-x}; shift
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/pmcx_$tag
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline $@"
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL --kernel-trace --output-format csv -d $OUT/a -o pmc -- $BENCH > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_INSTS_VSKIPPED SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC --kernel-trace --output-format csv -d $OUT/b -o pmc -- $BENCH > $OUT/b.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_INT64 SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $OUT/c -o pmc -- $BENCH > $OUT/c.log 2>&1
python3 - <<PY
import csv, glob, collections
for d in ("a","b","c"):
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % d, recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
        for row in csv.DictReader(open(f)):
            k = row.get("Kernel_Name","?")[:50]
            acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
            cnt[(k,row["Counter_Name"])] += 1
        for k, c in acc.items():
            for name, v in c.items():
                print("%-50s %-28s per-dispatch %.6g  (n=%d)" % (k, name, v/cnt[(k,name)], cnt[(k,name)]))
PY
