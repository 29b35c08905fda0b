#!/bin/bash
# usage: tools/pmc_passes.sh <tag> <workload> "<counters pass 1>" "<counters pass 2>" ...
# Each pass is one rocprofv3 --pmc run of tools/prof_step.py; summary printed + kept in gpurun_out/pmc_<tag>/
set -o pipefail
tag=$1; wl=$2; shift 2
repo=$PWD; out=$repo/gpurun_out/pmc_$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for ctrs in "$@"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $ctrs --output-format csv -d $out/p$i -- python3 $repo/tools/prof_step.py --workload $wl --steps 3 > $out/p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $out/p$i.log; exit 1; }
done
python3 $repo/tools/collate_prof.py $out $out/summary
grep unproject $out/summary_pmc_summary.csv
