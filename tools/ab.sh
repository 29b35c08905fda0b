#!/bin/bash
# A/B kernel variants: tools/ab.sh name1 name2 ...   (build_dev/<name>.so; "main" = in-tree lib)
for v in "$@"; do
  if [ "$v" = main ]; then unset TEMPO_LIB; else export TEMPO_LIB=$PWD/build_dev/$v.so; fi
  r=$(timeout -k 10 120 python bench.py --steps 30 --warmup 5 --no-cpu-baseline $BENCH_ARGS 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel_ms'], d['repack_ms'], d['ms_per_step'], d['value'])") || exit 1
  echo "$v kernel_ms repack_ms step_ms gsamp: $r"
done
