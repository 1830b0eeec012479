#!/bin/bash
# The rocprofv3 --pmc passes of a round: per scene three runs of one step of bench.py (one pass per run, never together with a
# trace), summarised into pmc.json, which bench.py prices its kernels with.
# usage: scripts/pmc_passes.sh <tag> [scene...]   -> gpurun_out/<tag>/pmc.json (+ pmc_summary.txt); also copied to profiles/<tag>/
# on the box, so that bench lines made later in the same call use it.
#   pass a: issue / wait / occupancy counters + HBM-side reads (FETCH_SIZE, 3 of the 4 TCC slots) + the clock (GRBM_GUI_ACTIVE)
#   pass b: the f64 / integer instruction classes + HBM-side writes (WRITE_SIZE)
#   pass c: the f32 instruction classes, lane utilisation (SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU), memory instructions, L2 hits
set -e
tag=$1; shift
scenes=${@:-cornell knight blocks_ball globe two_spheres}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$tag
mkdir -p $O $R/profiles/$tag
cd /tmp; export TMPDIR=/tmp
export RT_ONE_LANE=1   # one lane at a time (batches cut as usual): a dispatch's duration and clock are its own
A="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY FETCH_SIZE GRBM_GUI_ACTIVE"
B="SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_IOPS WRITE_SIZE"
C="SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS TCC_HIT_sum TCC_MISS_sum"
for s in $scenes; do
  for p in a b c; do
    case $p in a) counters=$A;; b) counters=$B;; c) counters=$C;; esac
    rm -rf $O/pmc_${p}_$s
    timeout -s KILL 150 rocprofv3 --pmc $counters --output-format csv -d $O/pmc_${p}_$s -- python3 $R/bench.py --steps 1 --warmup 0 --scene $s --no-cpu-baseline --no-extra --no-launch-timing > $O/pmc_${p}_$s.json 2> $O/pmc_${p}_$s.log
    echo "pass $p $s done"
  done
done
python3 $R/scripts/make_pmc.py $O > $O/pmc.json
for d in $(ls -d $O/pmc_*/ | xargs -n1 basename); do echo "== $d"; python3 $R/scripts/pmc_summary.py $O/$d | grep -E "k_bounce|k_extend|k_shade|k_raygen|k_accumulate|k_scan" || true; done > $O/pmc_summary.txt
cp $O/pmc.json $R/profiles/$tag/pmc.json
rm -rf $O/pmc_*/ $O/pmc_*.log
