#!/bin/bash
# A/B of library variants (build_dev/<name>.so; main = in-tree): root step repacked (chunked) and nhwc
# usage: tools/ab_lib.sh "<variants>" ["<extra bench flags>" ...]
vars=$1; shift
[ $# -eq 0 ] && set -- "" "--input-layout nhwc"
for v in $vars; do
  if [ "$v" = main ]; then unset TEMPO_LIB; else export TEMPO_LIB=$PWD/build_dev/$v.so; fi
  for args in "$@"; do
    r=$(timeout -k 10 120 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-module $args 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel_ms'], d['repack_ms'], d['ms_per_step'], d['value'], d['oracle_check']['max_abs_err'])") || exit 1
    echo "$v [$args] kernel_ms repack_ms step_ms gsamp err: $r"
  done
done
