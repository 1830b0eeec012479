#!/bin/bash
# usage: scripts/ab_env.sh "<bench args>" "VAR=a" "VAR=b" ...   (one bench run per environment setting)
args="$1"; shift
for setting in "$@"; do
  echo "== $setting $args"
  env $setting timeout -k 10 120 python bench.py --no-cpu-baseline $args | python -c "import json,sys; d=json.load(sys.stdin); print(round(d['value'],1), round(d['ms_per_step'],2), {k: round(v,2) for k,v in d['stage_ms_last_step'].items()})" || exit 1
done
