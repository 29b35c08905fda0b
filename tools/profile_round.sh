#!/bin/bash
# Run on the GPU box (through gpurun): kernel trace of bench.py + PMC passes of a short driver.
# usage: tools/profile_round.sh <tag> [workload] [extra bench flags]   -> gpurun_out/prof_<tag>/
set -o pipefail
tag=$1; wl=${2:-root}; shift; shift; extra="$*"
repo=$PWD
out=$repo/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -- python3 $repo/bench.py --steps 20 --warmup 5 --workload $wl $extra --no-cpu-baseline > $out/kt.log 2>&1 || { tail -5 $out/kt.log; exit 1; }
grep '"metric"' $out/kt.log > $out/bench_line.json
i=0
for ctrs in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_GATE_EN1_sum TA_TA_BUSY_sum TD_TD_BUSY_sum" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE TCP_PENDING_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $ctrs --output-format csv -d $out/p$i -- python3 $repo/tools/prof_step.py --workload $wl $extra --steps 3 > $out/p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $out/p$i.log; exit 1; }
done
python3 $repo/tools/collate_prof.py $out $out/summary
cat $out/summary_pmc_summary.csv; cat $out/bench_line.json
