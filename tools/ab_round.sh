#!/bin/bash
# GPU tests, then bench lines (run through gpurun from the repo root)
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/gputests.log 2>&1
tail -3 gpurun_out/gputests.log
rm -f gpurun_out/ab.log
for args in "" "--workload window" "--workload window --planes with" "--workload window --planes only" "--workload window_h36m --batch 128" "--workload window_h36m --batch 128 --planes only"; do
  echo "== $args" >> gpurun_out/ab.log
  timeout -k 10 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline $args >> gpurun_out/ab.log 2>&1 || break
done
cat gpurun_out/ab.log
