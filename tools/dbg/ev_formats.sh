R=${GRAFT_REPO_ROOT:-$PWD}; G=$R/gpurun_out; tag=${1:-r02l}
python $R/tools/bench_formats.py --size 2048 --steps 3 --qualities 0,1,2,3,4 2>/dev/null | grep format > $G/${tag}_formats_2048.jsonl
bash $R/tools/fuzz_all.sh 300 13 > $G/${tag}_fuzz.txt 2>&1
python $R/bench.py > $G/${tag}_bench.json 2> $G/${tag}_bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $G/prof_${tag}_formats -o trace -- python $R/tools/bench_formats.py --size 2048 --steps 3 --qualities 2,3 > /dev/null 2>&1
for f in $(find $G/prof_${tag}_formats -name "*kernel_stats.csv"); do cp $f $G/${tag}_formats_kernel_stats.csv; done
grep -c "0 mismatching" $G/${tag}_fuzz.txt; tail -c 300 $G/${tag}_bench.json
