#!/bin/bash
# The round's bench lines only (scripts/profile_round.sh without the rocprofv3 passes): run after profiles/<tag>/valu.json is in place,
# since bench.py takes the instructions per ray of the roofline from it.
set -e
tag=${1:-r02}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$tag
mkdir -p $O
B="python3 $R/bench.py --no-extra"
timeout -k 10 300 python3 $R/bench.py --steps 5 --warmup 1 > $O/bench_default.json
for s in cornell knight blocks_ball; do
  timeout -k 10 300 $B --steps 5 --warmup 1 --scene $s > $O/bench_$s.json
  timeout -k 10 300 $B --steps 10 --warmup 2 --scene $s --spp 13 --no-cpu-baseline > $O/bench_${s}_13spp.json
done
timeout -k 10 300 $B --steps 5 --warmup 1 --scene globe > $O/bench_globe.json
timeout -k 10 300 $B --steps 5 --warmup 1 --scene two_spheres > $O/bench_two_spheres.json
timeout -k 10 500 $B --steps 3 --warmup 1 --scene knight --triangles 1000000 > $O/bench_knight_1m.json
