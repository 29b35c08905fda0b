#!/bin/bash
# window workloads: unproject_kernel<WINDOW> (TEMPO_WINDOW_BRICK=0) vs the LDS-staged brick kernel
set -o pipefail
mkdir -p gpurun_out/ab
for br in 0 1; do
for v in "--workload window" "--workload window_h36m --batch 128"; do
  tag=$(echo "$v" | tr -d ' -')_brick$br
  TEMPO_WINDOW_BRICK=$br timeout -k 10 300 python bench.py --no-cpu-baseline --steps 10 --warmup 3 --blocks 3 $v > gpurun_out/ab/$tag.json 2> gpurun_out/ab/$tag.err || { echo "FAILED $v"; tail -5 gpurun_out/ab/$tag.err; exit 1; }
  python - "$tag" <<'PY'
import json,sys
d=json.load(open('gpurun_out/ab/%s.json'%sys.argv[1]))
r=d['roofline']
print(sys.argv[1], 'ms/step',d['ms_per_step'],'Gsamp/s',d['value'],'kernel_ms',r['kernel_ms'],'frac',r['frac'],'repack',d['repack_ms'],'err',d.get('oracle_check',{}).get('max_abs_err'), d['timing']['block_ms_per_step'])
PY
done
done
