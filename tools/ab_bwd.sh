#!/bin/bash
# A/B backward variants: tools/ab_bwd.sh name1 name2 ...  (build_dev/<name>.so; "main" = in-tree lib)
for v in "$@"; do
  if [ "$v" = main ]; then unset TEMPO_LIB; else export TEMPO_LIB=$PWD/build_dev/$v.so; fi
  r=$(timeout -k 10 200 python bench.py --workload ${WL:-window} --backward --steps 3 --warmup 1 --no-cpu-baseline $BENCH_ARGS 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['training_step'])") || exit 1
  echo "$v: $r"
done
