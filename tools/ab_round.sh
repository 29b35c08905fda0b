#!/bin/bash
# GPU tests (run through gpurun from the repo root)
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread -s > gpurun_out/gputests.log 2>&1
grep -E "grad max-abs|passed|failed|Error" gpurun_out/gputests.log | tail -20
