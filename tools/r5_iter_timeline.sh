#!/bin/bash
# kernel timeline of the last step of a short bench run: the coarse-to-fine start and one CG iteration
cd "$GRAFT_REPO_ROOT"; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5_iter; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-side 0 --no-accuracy --no-cold "$@" > $O/trace.log 2>&1
cd $R
python3 tools/trace_list.py $O/trace 60000 > $O/step_timeline.txt
rm -rf $O/trace
