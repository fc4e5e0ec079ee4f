#!/bin/bash
# one GPU iteration on the ASTC kernel (GPU box): parity tests of the product library, then A/B kernel timings
#   bash tools/dbg/astc_step.sh lib1.so lib2.so ...     (libraries under tools/ab/, timed against each other)
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out
(timeout 1500 python -m pytest $R/tests/test_gpu_astc.py $R/tests/test_gpu_real_blocks.py -x -q 2>&1 | tail -8) > $R/gpurun_out/astc_step_tests.log
tail -4 $R/gpurun_out/astc_step_tests.log
for rep in 1 2; do
for lib in "$@"; do
  echo "== rep $rep $lib"
  CFHIP_LIB=$R/$lib python $R/tools/bench_formats.py --size 2048 --steps 3 --formats ASTC_4x4,ASTC_6x6,ASTC_8x8 --qualities 2,3,4 2>/dev/null | grep format | python3 -c "
import sys, json
rows = list(map(json.loads, sys.stdin))
for f in sorted({r['format'] for r in rows}):
    print(f, '  '.join('%s/q%d %.3f' % (d['type'][:2], d['quality'], d['kernel_ms']) for d in rows if d['format'] == f))"
done
done 2>&1 | tee $R/gpurun_out/astc_step_bench.log
