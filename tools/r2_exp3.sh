#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2_exp3
mkdir -p $O
./tools/micro/dpp_test > $O/dpp.txt 2>&1
python -m pytest tests/test_gpu_operator.py tests/test_gpu_solve.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
NOSOLVE= python tools/exp_variants.py > $O/variants.log 2>&1
python bench.py --steps 5 --warmup 2 --cpu-side 0 > $O/bench.json 2>$O/bench.err
tail -3 $O/tests.log; cat $O/variants.log; cat $O/dpp.txt | cut -c1-200
