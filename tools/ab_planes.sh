#!/bin/bash
# A/B of library variants on the fused window planes: tools/ab_planes.sh name1 name2 ... (build_dev/<name>.so; main = in-tree)
for v in "$@"; do
  if [ "$v" = main ]; then unset TEMPO_LIB; else export TEMPO_LIB=$PWD/build_dev/$v.so; fi
  r=$(timeout -k 10 200 python bench.py --workload window --planes only --steps 10 --warmup 3 --no-cpu-baseline --no-module 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])") || exit 1
  echo "$v: $r"
done
