#!/bin/bash
# fused window planes: frames per chunk (repack + planes kernel per chunk)
for args in "--workload window" "--workload window_h36m --batch 128"; do
  for c in 0 2 4 8 16; do
    extra=""; [ $c -gt 0 ] && extra="--window-chunk $c"
    r=$(timeout -k 10 200 python bench.py $args --planes only $extra --steps 10 --warmup 3 --no-cpu-baseline --no-module 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])") || exit 1
    echo "$args chunk $c: $r"
  done
done
