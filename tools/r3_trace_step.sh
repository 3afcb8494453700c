#!/bin/bash
# kernel-by-kernel timeline of the last bench step (headline): tools/r3_trace_step.sh [bench args]
cd "$GRAFT_REPO_ROOT"; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3_trace_step; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-side 0 --no-accuracy --no-cold "$@" > $O/trace.log 2>&1
cd $R
python3 tools/trace_list.py $O/trace 12000 > $O/list.txt
python3 tools/trace_gaps.py $O/trace 10 > $O/gaps.txt
rm -rf $O/trace
grep -c fillBuffer $O/list.txt; tail -3 $O/gaps.txt
