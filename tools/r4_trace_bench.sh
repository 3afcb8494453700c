#!/bin/bash
# kernel-by-kernel timeline of the last step of bench.py: tools/r4_trace_bench.sh <name> [bench args]
name=$1; shift
cd "$GRAFT_REPO_ROOT"; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$name; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-side 0 --no-accuracy --no-cold "$@" > $O/trace.log 2>&1
cd $R
python3 tools/trace_list.py $O/trace 400000 > $O/list.txt
python3 tools/trace_gaps.py $O/trace 30 > $O/gaps.txt
rm -rf $O/trace
tail -1 $O/trace.log | cut -c1-300; tail -2 $O/gaps.txt
