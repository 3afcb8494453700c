#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2_2d; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_multilevel.py tests/test_gpu_slabs.py tests/test_gpu_fullsize.py -x -q -m gpu > $O/tests.log 2>&1; tail -2 $O/tests.log
for env in "FI_DUMMY=1" "FI_NO_FUSED_SMOOTHER=1"; do for cfg in 2 3; do
env $env timeout -k 10 300 python bench.py --config $cfg --steps 3 --warmup 1 --cpu-side 0 --no-accuracy > $O/b.json 2> $O/b.err && python -c "
import json; d=json.loads(open('$O/b.json').read().strip().splitlines()[-1]); c=d['config']; print('[$env] config $cfg', round(d['ms_per_step'],2), c['iterations'], c['coarse_iterations'], round(c['assemble_ms'],2), round(c['solve_ms'],2))"
done; done
