#!/bin/bash
# Runs the instruction-class calibration (scripts/micro/valu_mix.hip, built as bin/valu_mix on the CPU box) on the GPU box:
# the timing table, then three rocprofv3 --pmc passes of the same program (one run each, never with a trace), summarised
# per class.  usage: scripts/valu_mix_pmc.sh <tag>   -> gpurun_out/<tag>/valu_mix.txt, valu_mix_counters.txt
set -e
tag=${1:-r03}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$tag
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout -k 10 120 $R/bin/valu_mix > $O/valu_mix.txt
A="SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT"
B="SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES"
C="SQ_INSTS_VALU_FLOPS_FP64 SQ_INSTS_VALU_FLOPS_FP32 SQ_INSTS_VALU_IOPS SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAVES SQ_WAIT_INST_ANY SQ_WAIT_ANY"
timeout -s KILL 90 rocprofv3 --pmc $A --output-format csv -d $O/vm_a -- $R/bin/valu_mix > $O/vm_a.log 2>&1
timeout -s KILL 90 rocprofv3 --pmc $B --output-format csv -d $O/vm_b -- $R/bin/valu_mix > $O/vm_b.log 2>&1
timeout -s KILL 90 rocprofv3 --pmc $C --output-format csv -d $O/vm_c -- $R/bin/valu_mix > $O/vm_c.log 2>&1
python3 $R/scripts/valu_mix_counters.py $O/vm_a $O/vm_b $O/vm_c > $O/valu_mix_counters.txt
rm -rf $O/vm_a $O/vm_b $O/vm_c
cat $O/valu_mix.txt; cat $O/valu_mix_counters.txt
