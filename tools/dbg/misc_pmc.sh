#!/bin/bash
# Instruction counters of the non-BC7 kernels at Normal (GPU box): bash tools/dbg/misc_pmc.sh > summary.txt
# (FETCH_SIZE / WRITE_SIZE are collected by tools/profile.sh only: passed to rocprofv3 like the SQ counters they abort it)
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/misc_pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for spec in "BC6H bc6h" "EAC_R11 etc" "EAC_R11G11 etc" "BC4 bc15" "BC3 bc15" "ETC2_R8G8B8 etc" "ASTC_6x6 astc"; do
  set -- $spec; fmt=$1; pat=$2
  CMD="python $R/tools/bench_formats.py --size 2048 --steps 2 --formats $fmt --qualities 2"
  echo "== $fmt Quality::Normal 2048x2048 (per dispatch)"
  for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES"; do
    tag=${fmt}_$(echo $grp | cut -d' ' -f1)
    timeout 120 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/$tag -o pmc -- $CMD > $OUT/$tag.log 2>&1
    python3 - <<PY
import csv, glob, collections
for f in glob.glob("$OUT/$tag/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name","?")
        if "$pat" not in k: continue
        k = k[:58]
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"]); cnt[(k,row["Counter_Name"])] += 1
    for k, c in sorted(acc.items()):
        print("  %-58s %s" % (k, "  ".join("%s %.5g" % (n, v/cnt[(k,n)]) for n, v in sorted(c.items()))))
PY
  done
done
