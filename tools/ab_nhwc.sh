#!/bin/bash
# nhwc in-place root step: library variants x chunk sizes (run through gpurun from the repo root)
# usage: tools/ab_nhwc.sh "<variants>" "<chunks>" [extra bench flags]
vars=$1; chunks=$2; shift 2
for v in $vars; do
  if [ "$v" = main ]; then unset TEMPO_LIB; else export TEMPO_LIB=$PWD/build_dev/$v.so; fi
  for c in $chunks; do
    r=$(timeout -k 10 120 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --input-layout nhwc --chunk $c "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel_ms'], d['ms_per_step'], d['value'], d['oracle_check']['max_abs_err'])") || exit 1
    echo "$v chunk=$c kernel_ms step_ms gsamp err: $r"
  done
done
