#!/bin/bash
# Round-end evidence pass on the GPU box: bash tools/final_evidence.sh <tag>
# profile passes (kernel trace + PMC, separate runs) for BC7 at Normal / High / Highest, the bench
# lines that quote them, the formats table, the BASELINE configs, C5, and the parity sweep.
tag=${1:-r02b}
R=${GRAFT_REPO_ROOT:-$PWD}
G=$R/gpurun_out
mkdir -p $G
for q in 2 3 4; do
  sfx=""; [ $q != 2 ] && sfx="_q$q"
  kern="cfhip_bc7_encode_kernel<0, true, false>"; mang="cfhip_bc7_encode_kernelILi0ELb1ELb0E"
  # (round 5: High runs the wide kernel too)
  [ $q -ge 3 ] && kern="cfhip_bc7_encode_kernel<0, true, true>" && mang="cfhip_bc7_encode_kernelILi0ELb1ELb1E"
  bash $R/tools/profile.sh ${tag}$sfx --quality $q > $G/${tag}${sfx}_bc7_pmc_summary.txt 2>&1
  python $R/tools/pmc_to_json.py $G/prof_${tag}$sfx --tag ${tag}$sfx --quality $q --kernel "$kern" --mangled "$mang" --out $G/bc7_pmc$sfx.json > /dev/null \
    && cp $G/bc7_pmc$sfx.json $R/profiles/bc7_pmc$sfx.json
  for f in $(find $G/prof_${tag}$sfx/stats -name "*kernel_stats.csv"); do cp $f $G/${tag}${sfx}_bc7_kernel_stats.csv; done
  extra=""; [ $q != 2 ] && extra="--no-cpu-baseline"
  python $R/bench.py --quality $q $extra > $G/${tag}${sfx}_bench.json 2> $G/${tag}${sfx}_bench.err
done
# the second tile (synth.photo2: the BC7 mode split of real photographs): kernel stats + PMC passes of BC7 Normal on it
PROFILE_CMD="python $R/tools/bench_formats.py --size 4096 --steps 3 --formats BC7 --qualities 2 --tile photo2" \
  bash $R/tools/profile.sh ${tag}_photo2 > $G/${tag}_bc7_photo2_pmc_summary.txt 2>&1
for f in $(find $G/prof_${tag}_photo2/stats -name "*kernel_stats.csv"); do cp $f $G/${tag}_bc7_photo2_kernel_stats.csv; done
[ -n "$ONLY_BC7" ] && { tail -c 600 $G/${tag}_bench.json; exit 0; }
# ASTC at BASELINE config 3 (6x6 High, 4096x4096): kernel stats + PMC passes incl. FETCH_SIZE / WRITE_SIZE
PROFILE_CMD="python $R/tools/bench_formats.py --size 4096 --steps 3 --formats ASTC_6x6 --qualities 3" \
  bash $R/tools/profile.sh ${tag}_astc > $G/${tag}_astc_pmc_summary.txt 2>&1
for f in $(find $G/prof_${tag}_astc/stats -name "*kernel_stats.csv"); do cp $f $G/${tag}_astc_kernel_stats.csv; done
bash $R/tools/dbg/astc_dense_ab.sh > $G/${tag}_astc_dense_ab.txt 2>&1
# what bounds that kernel (VALU busy share, LDS bank-conflict share) -> profiles/astc_c3_pmc.json, replayed by `bench.py --config c3`
bash $R/tools/dbg/astc_lds_pmc.sh > $G/${tag}_astc_lds_pmc.txt 2>&1
( cd $R && python tools/astc_pmc_to_json.py $G/${tag}_astc_pmc_summary.txt $G/${tag}_astc_lds_pmc.txt $tag > /dev/null && cp profiles/astc_c3_pmc.json $G/astc_c3_pmc.json )
python $R/bench.py --config c3 > $G/${tag}_c3_bench.json 2> $G/${tag}_c3_bench.err
# ETC2 RGB at Normal, 2048x2048: SALU / VALU counts and traffic
PROFILE_CMD="python $R/tools/bench_formats.py --size 2048 --steps 3 --formats ETC2_R8G8B8 --qualities 2" \
  bash $R/tools/profile.sh ${tag}_misc > $G/${tag}_misc_pmc_summary.txt 2>&1
# the N-rank flow as the driver launches it (plain `python bench.py --gpus 2`); one device: gloo hook
BENCH_DIST_BACKEND=gloo python $R/bench.py --gpus 2 --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | grep "^{" > $G/${tag}_bench_gpus2_gloo.json
BENCH_DIST_BACKEND=gloo python $R/bench.py --config c5 --gpus 2 --textures 16 2>/dev/null | grep "^{" > $G/${tag}_c5_16tex_gpus2_gloo.json
python $R/tools/bench_formats.py --size 2048 --steps 3 --qualities 0,1,2,3,4 2>/dev/null | grep format > $G/${tag}_formats_2048.jsonl
python $R/tools/bench_formats.py --size 2048 --steps 3 --qualities 0,1,2,3,4 --tile photo2 2>/dev/null | grep format > $G/${tag}_formats_2048_photo2.jsonl
python $R/tools/bench_configs.py 2>/dev/null | grep "^{" > $G/${tag}_baseline_configs.jsonl
python $R/tools/bc1_bound.py --blocks 512 > $G/${tag}_bc1_bound.md 2>/dev/null
python $R/bench.py --config c5 --textures 16 2>/dev/null | grep "^{" > $G/${tag}_c5_16tex.json
python $R/tools/bench_srgb.py 2>/dev/null | grep "^{" > $G/${tag}_srgb_vs_linear.jsonl
bash $R/tools/fuzz_all.sh 300 7 > $G/${tag}_fuzz.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $G/prof_${tag}_formats -o trace -- python $R/tools/bench_formats.py --size 2048 --steps 3 --qualities 2,3 > /dev/null 2>&1
for f in $(find $G/prof_${tag}_formats -name "*kernel_stats.csv"); do cp $f $G/${tag}_formats_kernel_stats.csv; done
tail -c 1500 $G/${tag}_bench.json; echo; grep -c "0 mismatching" $G/${tag}_fuzz.txt
