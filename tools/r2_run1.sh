#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2_run1
mkdir -p $O
timeout -k 10 300 python bench.py --steps 5 --warmup 2 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
timeout -k 10 600 python -m pytest tests/test_gpu_two_ranks.py -x -q -m gpu > $O/two.log 2>&1; echo "two rc=$?"
tail -30 $O/two.log
cat $O/bench.json | cut -c1-2500
tail -3 $O/bench.err
