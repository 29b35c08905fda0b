set -o pipefail
repo=$PWD; out=$repo/gpurun_out/prof_sq; mkdir -p $out; cd /tmp; export TMPDIR=/tmp
i=0
for ctrs in "SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES" "SQ_INSTS_SMEM SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_BUSY_CU_CYCLES"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $ctrs --output-format csv -d $out/p$i -- python3 $repo/tools/prof_step.py --workload root --steps 3 > $out/p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $out/p$i.log; exit 1; }
done
python3 $repo/tools/collate_prof.py $out $out/summary
grep unproject_kernel $out/summary_pmc_summary.csv
