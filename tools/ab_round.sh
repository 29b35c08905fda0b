#!/bin/bash
# GPU tests, then the training-step timing (run through gpurun from the repo root)
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread -s > gpurun_out/gputests.log 2>&1
grep -E "grad max-abs|passed|failed|Error" gpurun_out/gputests.log | tail -20
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --backward > gpurun_out/bwd_root.log 2>&1 && timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --backward --workload window > gpurun_out/bwd_win.log 2>&1
grep -o "training_step.*" gpurun_out/bwd_root.log gpurun_out/bwd_win.log
