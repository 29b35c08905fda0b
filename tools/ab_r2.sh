#!/bin/bash
# round-2 A/B of the root step: projection shared vs per frame, NCHW (repack) vs NHWC in place
set -o pipefail
mkdir -p gpurun_out/ab
for v in "--projection shared" "--projection per-frame" "--projection shared --input-layout nhwc" "--projection per-frame --input-layout nhwc"; do
  tag=$(echo "$v" | tr -d ' -')
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-module $v > gpurun_out/ab/$tag.json 2> gpurun_out/ab/$tag.err || { echo "FAILED $v"; tail -5 gpurun_out/ab/$tag.err; exit 1; }
  python - "$tag" <<'PY'
import json,sys
d=json.load(open('gpurun_out/ab/%s.json'%sys.argv[1]))
r=d['roofline']
print(sys.argv[1], 'ms/step',d['ms_per_step'],'Gsamp/s',d['value'],'kernel_ms',r['kernel_ms'],'frac',r['frac'],'step_frac',r['step_frac'],'repack',d['repack_ms'],'table',d.get('footprint_table_ms'),'err',d.get('oracle_check',{}).get('max_abs_err'), d['timing']['block_ms_per_step'])
PY
done
