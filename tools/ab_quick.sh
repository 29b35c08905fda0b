#!/bin/bash
# quick A/B of the in-tree library (or TEMPO_LIB): step / kernel times of the main workloads
for args in "" "--input-layout nhwc" "--workload window" "--workload window --input-layout nhwc" "--workload window_h36m --batch 128" "--workload window_h36m --batch 128 --input-layout nhwc"; do
  r=$(timeout -k 10 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-module $args 2>gpurun_out/ab_quick.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel_ms'], d['repack_ms'], d['ms_per_step'], d['value'], d.get('oracle_check',{}).get('max_abs_err'))") || { tail -5 gpurun_out/ab_quick.err; exit 1; }
  echo "[$args] kernel_ms repack_ms step_ms gsamp err: $r"
done
