#!/bin/bash
# usage: scripts/pmc_ab.sh "<bench args>" "<counters>" lib...   -> per-kernel counter sums for each library
args="$1"; ctrs="$2"; shift; shift
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for lib in "$@"; do
  name=$(basename $lib .so)
  RT_MI355X_LIB=$lib timeout -s KILL 120 rocprofv3 --pmc $ctrs --output-format csv -d $R/gpurun_out/pmcab_$name -- python3 $R/bench.py --no-cpu-baseline $args > $R/gpurun_out/pmcab_$name.log 2>&1 || { echo "pmc run failed for $lib"; tail -5 $R/gpurun_out/pmcab_$name.log; exit 1; }
  echo "== $name"; python3 $R/scripts/pmc_summary.py $R/gpurun_out/pmcab_$name | grep -E "k_extend|k_shade|k_bounce"
done
