#!/bin/bash
# one GPU iteration on the ASTC kernel: parity tests, then kernel-only timings against a baseline library
#   bash tools/dbg/astc_iter.sh [base.so]
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out
(timeout 900 python -m pytest $R/tests/test_gpu_astc.py $R/tests/test_gpu_astc_hdr.py -x -q 2>&1 | tail -15) > $R/gpurun_out/astc_iter_tests.log
tail -3 $R/gpurun_out/astc_iter_tests.log
for lib in "$@" cuttlefish_amd/libcuttlefish_hip.so; do
  echo "== $lib"
  CFHIP_LIB=$R/$lib python $R/tools/bench_formats.py --size 2048 --steps 3 --formats ASTC_4x4,ASTC_6x6,ASTC_8x8,ASTC_12x12 --qualities 0,2,3,4 2>/dev/null | grep format | python3 -c "
import sys, json
rows = list(map(json.loads, sys.stdin))
for f in sorted({r['format'] for r in rows}):
    print(f, '  '.join('%s/q%d %.3f' % (d['type'][:2], d['quality'], d['kernel_ms']) for d in rows if d['format'] == f))"
done 2>&1 | tee $R/gpurun_out/astc_iter_bench.log
