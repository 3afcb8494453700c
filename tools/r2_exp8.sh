#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=$GRAFT_REPO_ROOT/gpurun_out/r2_exp8
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for v in base xpnosum; do
  if [ $v != base ]; then export FI_HIP_LIB=$GRAFT_REPO_ROOT/exp_libs/libfi_$v.so; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$v -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --cpu-side 0 --poly 4 --poly-ratio 30 > $O/prof_$v.log 2>&1
  python3 $GRAFT_REPO_ROOT/tools/trace_by_grid.py $O/prof_$v | grep -E "pcg_xp|pcg_resid|kernel time"
done
