#!/bin/bash
# tools/r4_trace_asm.sh <name> [ENV=..]: kernel list of one fi_assemble alone (tools/r4_asm_only.py) -> gpurun_out/<name>/list.txt
name=$1; shift
cd "$GRAFT_REPO_ROOT"; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$name; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
python3 tools/r4_asm_only.py | tail -1
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 $R/tools/r4_asm_only.py > $O/trace.log 2>&1
cd $R
python3 tools/trace_list.py $O/trace 40000 > $O/list.txt
rm -rf $O/trace
tail -1 $O/trace.log
