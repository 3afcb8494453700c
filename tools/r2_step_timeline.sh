#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=$GRAFT_REPO_ROOT/gpurun_out/r2_exp13
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --cpu-side 0 --no-accuracy > $O/trace.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/trace_gaps.py $O/trace 10 > $O/gaps.txt
python3 $GRAFT_REPO_ROOT/tools/trace_list.py $O/trace 20000 > $O/list.txt
python3 $GRAFT_REPO_ROOT/tools/trace_by_grid.py $O/trace > $O/by_grid.md
rm -rf $O/trace
