#!/bin/bash
# kernel statistics of a short bench run: names matching $1 (egrep)
cd "$GRAFT_REPO_ROOT"; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5_kstat; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
pat=$1; shift
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-side 0 --no-accuracy --no-cold "$@" > $O/trace.log 2>&1
cd $R
python3 tools/trace_by_grid.py $O/trace > $O/by_grid.md
rm -rf $O/trace
grep -E "$pat" $O/by_grid.md | cut -c1-160
