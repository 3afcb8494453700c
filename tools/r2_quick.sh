#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2_quick; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_operator.py tests/test_gpu_solve.py -x -q -m gpu > $O/tests.log 2>&1; tail -2 $O/tests.log
for env in "" "FI_SERIAL_LEVELS=1"; do
env $env timeout -k 10 300 python bench.py --steps 5 --warmup 2 --cpu-side 0 --no-accuracy > $O/bench.json 2> $O/bench.err; python -c "
import json; d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); c=d['config']; print('[$env]', round(d['ms_per_step'],3), '%.4g' % d['value'], c['iterations'], c['coarse_iterations'], round(c['assemble_ms'],2), round(c['solve_ms'],2))"
done
