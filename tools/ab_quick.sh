#!/bin/bash
# quick A/B of the in-tree library: root step, repacked (chunked) and nhwc in place (one launch)
for args in "" "--input-layout nhwc --chunk 32" "--workload window" "--workload window_h36m --batch 128"; do
  r=$(timeout -k 10 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline $args 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel_ms'], d['repack_ms'], d['ms_per_step'], d['value'], d.get('oracle_check',{}).get('max_abs_err'))") || exit 1
  echo "[$args] kernel_ms repack_ms step_ms gsamp err: $r"
done
