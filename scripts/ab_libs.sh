#!/bin/bash
# usage (on the GPU box): scripts/ab_libs.sh "<bench args>" lib1.so lib2.so ...   one bench line per library: value, ms/step, stage ms
args="$1"; shift
for lib in "$@"; do
  RT_MI355X_LIB=$PWD/$lib timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-extra $args > /tmp/ab_line.json 2>/tmp/ab_line.err || { tail -5 /tmp/ab_line.err; exit 1; }
  python3 -c "import json,sys; d=json.load(open('/tmp/ab_line.json')); print('$lib', '$args', round(d['value'],1), round(d['ms_per_step'],2), {k: round(v,2) for k,v in d['stage_ms_last_step'].items() if k in ('extend','shade','tail')})"
done
