#!/bin/bash
# kernel-by-kernel timeline of the last step of tools/r4_acc.py (the headline leg): tools/r4_trace.sh <name> [ENV=.. ...]
name=$1; shift
cd "$GRAFT_REPO_ROOT"; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$name; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
export REPS=2
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 $R/tools/r4_acc.py > $O/trace.log 2>&1
cd $R
python3 tools/trace_list.py $O/trace 40000 > $O/list.txt
python3 tools/trace_gaps.py $O/trace 15 > $O/gaps.txt
rm -rf $O/trace
tail -2 $O/trace.log; tail -3 $O/gaps.txt
