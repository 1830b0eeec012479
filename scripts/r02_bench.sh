#!/bin/bash
# Bench lines at 100 and 13 spp (a rank's share of a 600 px / 100 spp image on 8 GPUs) for the three scenes, and a
# kernel timeline of one 13-spp step.  usage: scripts/r02_bench.sh <tag> [scenes...]   (environment passes through)
set -e
tag=$1; shift
scenes=${@:-cornell knight blocks_ball}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$tag
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for s in $scenes; do
  timeout -k 10 200 python3 $R/bench.py --steps 5 --warmup 1 --scene $s --no-cpu-baseline --no-extra > $O/bench_${s}_100.json
  timeout -k 10 200 python3 $R/bench.py --steps 10 --warmup 2 --scene $s --spp 13 --no-cpu-baseline --no-extra > $O/bench_${s}_13.json
  if [ -z "$NO_TRACE" ]; then
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O/trace_${s}_13 -- python3 $R/bench.py --steps 2 --warmup 1 --scene $s --spp 13 --no-cpu-baseline --no-extra > $O/trace_${s}_13.log 2>&1
  python3 $R/scripts/timeline.py $O/trace_${s}_13 > $O/timeline_${s}_13.txt
  rm -rf $O/trace_${s}_13
  fi
  echo "== $s"; python3 -c "
import json,sys
for n in (100,13):
    d=json.load(open('$O/bench_${s}_%d.json'%n)); print(n, round(d['ms_per_step'],2), round(d['value']), d['stage_ms_last_step'])"
  if [ -z "$NO_TRACE" ]; then tail -12 $O/timeline_${s}_13.txt; fi
done
