#!/bin/bash
# usage (on the GPU box): scripts/ab_env_lib.sh <lib.so> "<bench args>" "VAR=a" "VAR=b VAR2=c" ...   one bench line per environment setting, with one library
lib=$1; args="$2"; shift; shift
for setting in "$@"; do
  env RT_MI355X_LIB=$PWD/$lib $setting timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-extra $args > /tmp/ab_line.json 2>/tmp/ab_line.err || { tail -5 /tmp/ab_line.err; exit 1; }
  python3 -c "import json,sys; d=json.load(open('/tmp/ab_line.json')); print('$lib', '$setting', '$args', round(d['value'],1), round(d['ms_per_step'],2), {k: round(v,2) for k,v in d['stage_ms_last_step'].items() if k in ('extend','shade','tail')})"
done
