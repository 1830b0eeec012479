#!/bin/bash
# Collects the round's evidence on the GPU box: bench lines, rocprofv3 kernel stats, PMC traffic.
# usage: scripts/profile_round.sh <tag>     (writes under gpurun_out/<tag>/)
set -e
tag=${1:-r01}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$tag
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 python3 $R/bench.py --steps 5 --warmup 1 > $O/bench_cornell.json
timeout -k 10 300 python3 $R/bench.py --steps 3 --warmup 1 --scene knight > $O/bench_knight.json
timeout -k 10 300 python3 $R/bench.py --steps 3 --warmup 1 --scene blocks_ball > $O/bench_blocks_ball.json
timeout -k 10 300 python3 $R/bench.py --steps 3 --warmup 1 --scene globe --no-cpu-baseline > $O/bench_globe.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_cornell -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $O/stats_cornell.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_knight -- python3 $R/bench.py --steps 3 --warmup 1 --scene knight --no-cpu-baseline > $O/stats_knight.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_cornell -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $O/pmc_fetch_cornell.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_cornell -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $O/pmc_write_cornell.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_knight -- python3 $R/bench.py --steps 1 --warmup 0 --scene knight --no-cpu-baseline > $O/pmc_fetch_knight.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_knight -- python3 $R/bench.py --steps 1 --warmup 0 --scene knight --no-cpu-baseline > $O/pmc_write_knight.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_blocks_ball -- python3 $R/bench.py --steps 1 --warmup 0 --scene blocks_ball --no-cpu-baseline > $O/pmc_fetch_blocks_ball.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_blocks_ball -- python3 $R/bench.py --steps 1 --warmup 0 --scene blocks_ball --no-cpu-baseline > $O/pmc_write_blocks_ball.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $O/pmc_sq_cornell -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $O/pmc_sq_cornell.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $O/pmc_sq_knight -- python3 $R/bench.py --steps 1 --warmup 0 --scene knight --no-cpu-baseline > $O/pmc_sq_knight.log 2>&1
for d in pmc_fetch_cornell pmc_write_cornell pmc_fetch_knight pmc_write_knight pmc_fetch_blocks_ball pmc_write_blocks_ball pmc_sq_cornell pmc_sq_knight; do echo "== $d"; python3 $R/scripts/pmc_summary.py $O/$d | grep -E "k_bounce|k_extend|k_shade|k_raygen|k_accumulate"; done > $O/pmc_summary.txt
python3 $R/scripts/make_traffic.py $O > $O/traffic.json
cat $O/bench_cornell.json $O/bench_knight.json $O/bench_blocks_ball.json; cat $O/pmc_summary.txt
