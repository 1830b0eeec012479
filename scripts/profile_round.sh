#!/bin/bash
# Collects a round's evidence on the GPU box.  usage: scripts/profile_round.sh <tag> <part>...   parts, in this order:
#   mix    instruction-class calibration (bin/valu_mix: timing table + three --pmc passes)   -> valu_mix.txt, valu_mix_counters.txt, valu_mix.json
#   pmc    rocprofv3 --pmc passes of one step per scene (scripts/pmc_passes.sh)               -> pmc.json, pmc_summary.txt
#   bench  the bench lines (they price their kernels with the pmc.json / valu_mix.json made above) -> bench_*.json
#   stats  rocprofv3 --kernel-trace --stats of the bench command, one lane and two lanes      -> *_kernel_stats.csv, timeline_*.txt
# Everything lands under gpurun_out/<tag>/ (and pmc.json / valu_mix.json also under profiles/<tag>/ of the box's copy of the tree,
# where bench.py looks); copy what is to be judged into profiles/<tag>/ afterwards.  PMC passes are runs of their own, never
# together with a trace.
set -e
tag=$1; shift
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$tag
mkdir -p $O $R/profiles/$tag
cd /tmp; export TMPDIR=/tmp
B="python3 $R/bench.py --no-extra"
for part in "$@"; do
case $part in
mix)
  bash $R/scripts/valu_mix_pmc.sh $tag > /dev/null
  python3 $R/scripts/make_valu_mix.py $O/valu_mix.txt > $O/valu_mix.json
  cp $O/valu_mix.json $R/profiles/$tag/valu_mix.json
  ;;
pmc)
  bash $R/scripts/pmc_passes.sh $tag cornell knight blocks_ball globe two_spheres
  ;;
bench)
  timeout -k 10 300 python3 $R/bench.py --steps 5 --warmup 1 > $O/bench_default.json
  for s in cornell knight blocks_ball globe two_spheres; do
    timeout -k 10 300 $B --steps 5 --warmup 1 --scene $s > $O/bench_$s.json
  done
  for s in cornell knight blocks_ball; do
    for spp in 50 25 13 1; do
      timeout -k 10 300 $B --steps 10 --warmup 2 --scene $s --spp $spp --no-cpu-baseline > $O/bench_${s}_${spp}spp.json
    done
  done
  timeout -k 10 500 $B --steps 3 --warmup 1 --scene knight --triangles 1000000 > $O/bench_knight_1m.json
  # the fast mode (same kernels in f32), its throughput next to its measured ID-mismatch rate (scripts/measure_f32.py)
  for s in cornell knight blocks_ball; do
    timeout -k 10 300 $B --steps 5 --warmup 1 --scene $s --precision f32 --no-cpu-baseline > $O/bench_${s}_f32.json
  done
  timeout -k 10 300 python3 $R/scripts/measure_f32.py > $O/f32_mode.json
  ;;
stats)
  for s in cornell knight blocks_ball globe two_spheres; do
    # (RT_ONE_LANE=1: one lane at a time, batches cut as usual - per-launch durations that are the kernel's own, as in the
    # instrumented step of the bench line; the default two lanes stretch them, see *_2lanes_kernel_stats.csv)
    RT_ONE_LANE=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats1_$s -- $B --steps 5 --warmup 1 --scene $s --no-cpu-baseline --no-launch-timing > $O/stats_$s.log 2>&1
    cp $(find $O/stats1_$s -name "*kernel_stats.csv" | head -1) $O/${s}_f64_kernel_stats.csv
    rm -rf $O/stats1_$s
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$s -- $B --steps 5 --warmup 1 --scene $s --no-cpu-baseline --no-launch-timing > $O/stats_$s.log 2>&1
    cp $(find $O/stats_$s -name "*kernel_stats.csv" | head -1) $O/${s}_f64_2lanes_kernel_stats.csv
    python3 $R/scripts/timeline.py $O/stats_$s 200 > $O/timeline_${s}.txt || true
    rm -rf $O/stats_$s $O/stats_$s.log
  done
  ;;
esac
echo "part $part done"
done
