#!/bin/bash
# miss-latency / L2 / TA counters of the nhwc 32-frame launch for library variants
# usage: tools/pmc_lat.sh "<variants>"   (main = in-tree lib)
set -o pipefail
repo=$PWD
cd /tmp && export TMPDIR=/tmp
for v in $1; do
  if [ "$v" = main ]; then unset TEMPO_LIB; else export TEMPO_LIB=$repo/build_dev/$v.so; fi
  out=$repo/gpurun_out/pmc_lat_$v; rm -rf $out; mkdir -p $out
  i=0
  for ctrs in "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TA_TA_BUSY_sum GRBM_GUI_ACTIVE" "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum"; do
    i=$((i+1))
    timeout -s KILL 120 rocprofv3 --pmc $ctrs --output-format csv -d $out/p$i -- python3 $repo/tools/prof_step.py --workload root --input-layout nhwc --chunk 32 --steps 3 > $out/p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $out/p$i.log; exit 1; }
  done
  python3 $repo/tools/collate_prof.py $out $out/summary
  echo "== $v"; grep unproject_kernel $out/summary_pmc_summary.csv
done
