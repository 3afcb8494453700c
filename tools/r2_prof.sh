#!/bin/bash
# rocprofv3 kernel trace of the bench step: usage r2_prof.sh <name> [bench args...]
cd "$GRAFT_REPO_ROOT"
name=$1; shift
O=$GRAFT_REPO_ROOT/gpurun_out/$name
mkdir -p $O
python bench.py --steps 5 --warmup 2 --cpu-side 0 "$@" > $O/bench.json 2> $O/bench.err
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --cpu-side 0 "$@" > $O/prof.log 2>&1
cd "$GRAFT_REPO_ROOT"
python tools/trace_by_grid.py $O/prof > $O/by_grid.md
cat $O/bench.json | cut -c1-700; cat $O/by_grid.md
