#!/bin/bash
# Instruction counts of the hot kernels (rocprofv3 --pmc, one pass per counter set; kernel trace only).
cd "$GRAFT_REPO_ROOT"; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5_pmc_insts; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
cd /tmp
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/p$i -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-side 0 --no-accuracy --no-cold --datasets 1 > $O/run$i.log 2>&1 || { echo "pass $i failed"; tail -5 $O/run$i.log; }
done
cd $R
python3 tools/r5_pmc_insts.py $O > $O/summary.txt
rm -rf $O/p*
cat $O/summary.txt
