#!/bin/bash
# GPU tests, then the root/window A/B lines (run through gpurun from the repo root)
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/gputests.log 2>&1
tail -5 gpurun_out/gputests.log
BENCH_ARGS="--no-share" tools/ab.sh main abl1 abl4 abl6 > gpurun_out/ab_root.log 2>&1 &&
tools/ab.sh main > gpurun_out/ab_share.log 2>&1 &&
BENCH_ARGS="--group-max 4" tools/ab.sh main >> gpurun_out/ab_share.log 2>&1 &&
BENCH_ARGS="--group-max 16" tools/ab.sh main >> gpurun_out/ab_share.log 2>&1 &&
BENCH_ARGS="--group-max 32" tools/ab.sh main >> gpurun_out/ab_share.log 2>&1 &&
BENCH_ARGS="--workload window" tools/ab.sh main abl1 abl4 abl6 > gpurun_out/ab_win.log 2>&1
cat gpurun_out/ab_root.log gpurun_out/ab_share.log gpurun_out/ab_win.log
