#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=$GRAFT_REPO_ROOT/gpurun_out/r2_c5; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 $GRAFT_REPO_ROOT/bench.py --config 5 --steps 1 --warmup 0 --cpu-side 0 --no-accuracy > $O/trace.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/trace_by_grid.py $O/trace > $O/by_grid.md
rm -rf $O/trace
head -45 $O/by_grid.md
