#!/bin/bash
# A/B of library variants on the training step through the planes:
# tools/ab_planes_train.sh name1 name2 ...   (build_dev/<name>.so; main = in-tree lib)
for wl in "--workload window" "--workload window_h36m --batch 128"; do
for v in "$@"; do
  if [ "$v" = main ]; then unset TEMPO_LIB; else export TEMPO_LIB=$PWD/build_dev/$v.so; fi
  r=$(timeout -k 10 200 python bench.py $wl --planes only --backward --steps 5 --warmup 2 --no-cpu-baseline --no-module 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); t=d['training_step']; print(t['planes_forward_ms'], t['claim_pass_ms'], t['backward_ms'])")
  echo "[$wl] $v planes claims backward: $r"
done
done
