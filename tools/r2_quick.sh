#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2_quick; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_poly.py tests/test_gpu_multilevel.py -x -q -m gpu > $O/tests.log 2>&1; tail -3 $O/tests.log
for args in "" "--side 512 --points 8000000" "--dtype f64"; do
timeout -k 10 300 python bench.py --steps 3 --warmup 1 --cpu-side 0 --no-accuracy $args > $O/bench.json 2> $O/bench.err; python -c "
import json; d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); c=d['config']; print('$args', round(d['ms_per_step'],3), '%.4g' % d['value'], c['iterations'], c['coarse_iterations'], round(c['assemble_ms'],2), round(c['solve_ms'],2), round(d['roofline']['frac'],3), round(d['roofline']['launch_ms']*1e3,1), round(d['roofline_apply']['frac'],3))"
done
