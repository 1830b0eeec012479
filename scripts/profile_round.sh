#!/bin/bash
# Collects the round's evidence on the GPU box: bench lines, rocprofv3 kernel stats, PMC traffic and instruction counts.
# usage: scripts/profile_round.sh <tag>     (writes under gpurun_out/<tag>/; copy what is to be judged into profiles/<tag>/)
# PMC passes: one rocprofv3 --pmc run each (FETCH_SIZE, WRITE_SIZE, the SQ set), never together with a trace.
set -e
tag=${1:-r02}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$tag
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
B="python3 $R/bench.py --no-extra"
echo "bench lines"
timeout -k 10 300 python3 $R/bench.py --steps 5 --warmup 1 > $O/bench_default.json
for s in cornell knight blocks_ball; do
  timeout -k 10 300 $B --steps 5 --warmup 1 --scene $s > $O/bench_$s.json
  timeout -k 10 300 $B --steps 10 --warmup 2 --scene $s --spp 13 --no-cpu-baseline > $O/bench_${s}_13spp.json
done
timeout -k 10 300 $B --steps 5 --warmup 1 --scene globe > $O/bench_globe.json
timeout -k 10 300 $B --steps 5 --warmup 1 --scene two_spheres > $O/bench_two_spheres.json
timeout -k 10 500 $B --steps 3 --warmup 1 --scene knight --triangles 1000000 > $O/bench_knight_1m.json
echo "kernel stats"
for s in cornell knight blocks_ball; do
  # (RT_ONE_LANE=1: one lane at a time, batches cut as usual - per-launch durations that are the kernel's own, as in the
  # timed step of the bench line; the default two lanes stretch them, see *_2lanes_kernel_stats.csv)
  RT_ONE_LANE=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats1_$s -- $B --steps 5 --warmup 1 --scene $s --no-cpu-baseline > $O/stats_$s.log 2>&1
  cp $(find $O/stats1_$s -name "*kernel_stats.csv" | head -1) $O/${s}_f64_kernel_stats.csv
  rm -rf $O/stats1_$s
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$s -- $B --steps 5 --warmup 1 --scene $s --no-cpu-baseline > $O/stats_$s.log 2>&1
  cp $(find $O/stats_$s -name "*kernel_stats.csv" | head -1) $O/${s}_f64_2lanes_kernel_stats.csv
  python3 $R/scripts/timeline.py $O/stats_$s 200 > $O/timeline_${s}.txt || true
  rm -rf $O/stats_$s
done
echo "pmc"
export RT_ONE_LANE=1
for s in cornell knight blocks_ball; do
  for c in fetch:FETCH_SIZE write:WRITE_SIZE "sq:SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY"; do
    name=${c%%:*}; counters=${c#*:}
    timeout -s KILL 150 rocprofv3 --pmc $counters --output-format csv -d $O/pmc_${name}_$s -- $B --steps 1 --warmup 0 --scene $s --no-cpu-baseline > $O/pmc_${name}_$s.log 2>&1
  done
done
for d in $(ls -d $O/pmc_*/ | xargs -n1 basename); do echo "== $d"; python3 $R/scripts/pmc_summary.py $O/$d | grep -E "k_bounce|k_extend|k_shade|k_raygen|k_accumulate" || true; done > $O/pmc_summary.txt
python3 $R/scripts/make_traffic.py $O > $O/traffic.json
python3 $R/scripts/make_valu.py $O > $O/valu.json
rm -rf $O/pmc_*/
cat $O/valu.json | head -40
