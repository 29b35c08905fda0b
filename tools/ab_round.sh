#!/bin/bash
# bench lines for every workload (run through gpurun from the repo root)
rm -f gpurun_out/ab.log
for args in "" "--input-layout nhwc" "--zmax with" "--zmax only" "--channels 32" "--channels 32 --input-layout nhwc" "--backward" "--workload window" "--workload window --input-layout nhwc" "--workload window --channels 32" "--workload window --channels 32 --planes only" "--workload window --backward --steps 5 --warmup 2" "--workload window --planes with" "--workload window --planes only" "--workload window --planes only --backward --steps 5 --warmup 2" "--workload window_h36m --batch 128" "--workload window_h36m --batch 128 --planes only" "--workload window_h36m --batch 128 --backward --steps 5 --warmup 2" "--workload window_h36m --batch 128 --planes only --backward --steps 5 --warmup 2" "--workload window --timesteps 4 --share" "--workload dense"; do
  echo "== $args" >> gpurun_out/ab.log
  timeout -k 10 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline $args 2>&1 | grep -v amdgpu.ids >> gpurun_out/ab.log || { echo FAILED >> gpurun_out/ab.log; break; }
done
python - <<'PY'
import json
for line in open('gpurun_out/ab.log'):
    if line.startswith('=='):
        print(line.strip()); continue
    try:
        d = json.loads(line)
    except Exception:
        print(line.strip()[:300]); continue
    r = d['roofline']
    print('  value %.2f Gsamp/s  step %.4f ms  kernel %.4f ms (%d launches)  repack %.4f  frac %.3f' % (
        d['value'], d['ms_per_step'], r['kernel_ms'], r['launches_per_step'], d['repack_ms'], r['frac']),
        {k: d[k] for k in ('unfused_consumer', 'training_step', 'zcolumns_ms', 'collective') if k in d})
PY
