#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2_jac; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_solve.py tests/test_gpu_slabs.py tests/test_gpu_multilevel.py -x -q -m gpu > $O/tests.log 2>&1; tail -2 $O/tests.log
for args in "--poly 0 --levels 2" "--config 5"; do
timeout -k 10 300 python bench.py --steps 3 --warmup 1 --cpu-side 0 --no-accuracy $args > $O/b.json 2> $O/b.err && python -c "
import json; d=json.loads(open('$O/b.json').read().strip().splitlines()[-1]); c=d['config']; print('$args', round(d['ms_per_step'],2), c['iterations'], c['coarse_iterations'], round(c['assemble_ms'],2), round(c['solve_ms'],2))"
done
