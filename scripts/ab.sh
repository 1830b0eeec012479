#!/bin/bash
# usage: scripts/ab.sh "<bench args>" lib1.so lib2.so ...   (prints value, ms/step, stage ms for each)
args="$1"; shift
for lib in "$@"; do
  echo "== $lib"
  RT_MI355X_LIB=$lib timeout -k 10 120 python bench.py --no-cpu-baseline --no-extra $args | python -c "import json,sys; d=json.load(sys.stdin); print(round(d['value'],1), round(d['ms_per_step'],2), {k: round(v,2) for k,v in d['stage_ms_last_step'].items()})" || exit 1
done
