#!/bin/bash
# usage: scripts/ab_env2.sh "<bench args>" "VAR=a" "VAR=b" ...   (one bench run per setting: value, ms per step, stages of the last step)
args="$1"; shift
for setting in "$@"; do
  echo -n "$setting: "
  env $setting timeout -k 10 120 python bench.py --no-cpu-baseline --no-extra $args | python -c "import json,sys; d=json.load(sys.stdin); s=d['stage_ms_last_step']; print(round(d['value'],1), round(d['ms_per_step'],2), 'extend', round(s['extend'],2), 'shade', round(s['shade'],2), 'tail', round(s['tail'],2))" || exit 1
done
