#!/bin/bash
# Short renders over many sample counts and sizes (every run under its own time limit; stops at the first failure).
for s in knight blocks_ball cornell blue_ball two_spheres globe; do
  for spp in 1 2 3 7 16 33 64 65 130; do
    echo -n "$s spp=$spp: "
    timeout -k 10 100 python bench.py --no-cpu-baseline --no-extra --scene $s --spp $spp --steps 2 --warmup 1 | python -c "import json,sys; d=json.load(sys.stdin); print(round(d['value'],1), round(d['ms_per_step'],2))" || exit 1
  done
done
for w in 8 64 200 333 1024; do
  echo -n "knight width=$w: "
  timeout -k 10 100 python bench.py --no-cpu-baseline --no-extra --scene knight --width $w --spp 20 --steps 2 --warmup 1 | python -c "import json,sys; d=json.load(sys.stdin); print(round(d['value'],1), round(d['ms_per_step'],2))" || exit 1
done
