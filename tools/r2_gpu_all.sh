#!/bin/bash
# Full GPU suite + the default bench line.
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2_all; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $O/tests.log 2>&1; tail -5 $O/tests.log
timeout -k 10 300 python bench.py --steps 5 --warmup 2 > $O/bench.json 2> $O/bench.err; tail -c 2500 $O/bench.json
