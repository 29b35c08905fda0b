#!/bin/bash
# Run on the GPU box (through gpurun): kernel trace + PMC passes of the backward launch.
# usage: tools/profile_bwd.sh <tag> [workload] [extra bench flags]   -> gpurun_out/prof_<tag>/
set -o pipefail
tag=$1; wl=${2:-window}; shift; shift; extra="$*"
repo=$PWD
out=$repo/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -- python3 $repo/tools/prof_bwd.py --workload $wl $extra --steps 5 > $out/kt.log 2>&1 || { tail -5 $out/kt.log; exit 1; }
i=0
for ctrs in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE TCC_ATOMIC_sum"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $ctrs --output-format csv -d $out/p$i -- python3 $repo/tools/prof_bwd.py --workload $wl $extra --steps 2 > $out/p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $out/p$i.log; exit 1; }
done
python3 $repo/tools/collate_prof.py $out $out/summary
cat $out/summary_pmc_summary.csv
