#!/bin/bash
name=$1; shift
cd "$GRAFT_REPO_ROOT"; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$name; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
export REPS=2
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 $R/tools/r4_acc.py > $O/trace.log 2>&1
cd $R
python3 tools/trace_streams.py $O/trace 4500 > $O/streams.txt
python3 tools/trace_list.py $O/trace 40000 > $O/list.txt
rm -rf $O/trace
tail -1 $O/trace.log; head -3 $O/streams.txt
