#!/bin/bash
# Regression A/B against the round-2 tree (build_dev/r2 = git archive of 15aeede, library built there):
# the same bench flags through both trees, alternating.  usage: tools/ab_r2.sh "<bench flags>" ...
[ $# -eq 0 ] && set -- "" "--workload window" "--workload window --planes only"
for args in "$@"; do
  for rep in 1 2; do
    for tree in . build_dev/r2; do
      r=$(cd $tree && timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-module $args 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['roofline']['kernel_ms'], d['repack_ms'], d['ms_per_step'], d['value'])") || exit 1
      echo "$tree [$args] kernel_ms repack_ms step_ms gsamp: $r"
    done
  done
done
