#!/bin/bash
# A/B of the group walk in the tail launch: RT_COOP (live lanes per wave at or below which groups answer), RT_TAIL (paths at
# which the tail launch takes over).  usage: scripts/exp_coop.sh <scene> "<coop values>" "<tail values>" "<spp values>"
scene=$1
for spp in ${4:-13 100}; do
for tail in ${3:-1048576}; do
for c in $2; do
  echo -n "$scene spp=$spp tail=$tail coop=$c: "
  RT_TAIL=$tail RT_COOP=$c timeout -k 10 100 python bench.py --no-cpu-baseline --no-extra --scene $scene --spp $spp --steps 8 --warmup 2 | python -c "import json,sys; d=json.load(sys.stdin); print(round(d['value'],1), round(d['ms_per_step'],3))" || exit 1
done; done; done
