#!/bin/bash
# usage: scripts/r02_sweep.sh VAR "v1 v2 ..." [scenes...]: bench at 100 and 13 spp for every value of an environment switch
set -e
var=$1; vals=$2; shift; shift
scenes=${@:-cornell knight blocks_ball}
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
for v in $vals; do
  export $var=$v
  for s in $scenes; do
    a=$(timeout -k 10 200 python3 $R/bench.py --steps 5 --warmup 1 --scene $s --no-cpu-baseline --no-extra | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],2))")
    b=$(timeout -k 10 200 python3 $R/bench.py --steps 10 --warmup 2 --scene $s --spp 13 --no-cpu-baseline --no-extra | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],2))")
    echo "$var=$v $s: 100 spp $a ms, 13 spp $b ms"
  done
done
