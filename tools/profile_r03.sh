#!/bin/bash
# Round-3 profile set, on the GPU box: kernel trace + PMC passes per workload (tools/profile_round.sh).
# usage: tools/profile_r03.sh [tags...]   -> gpurun_out/prof_<tag>/ ; copy the summaries into profiles/r03/
set -o pipefail
declare -A WL=( [root]="root" [root_zmax_only]="root --zmax only" [root_c32]="root --channels 32"
                [window]="window" [window_c32]="window --channels 32"
                [planes]="window --planes only" [planes_c32]="window --planes only --channels 32" )
tags=${@:-root root_zmax_only root_c32 window window_c32 planes}
for t in $tags; do
  echo "== $t: ${WL[$t]}"
  bash tools/profile_round.sh $t ${WL[$t]} > gpurun_out/prof_$t.out 2>&1 || { echo "$t failed"; tail -5 gpurun_out/prof_$t.out; exit 1; }
  tail -1 gpurun_out/prof_$t.out | cut -c1-300
done
