#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2_cubic; mkdir -p $O
for env in "" "FI_LINEAR_START=1"; do for args in "" "--side 512 --points 8000000" "--levels 2"; do
env $env timeout -k 10 200 python bench.py --steps 3 --warmup 1 --cpu-side 0 --no-accuracy $args > $O/b.json 2> $O/b.err && python -c "
import json; d=json.loads(open('$O/b.json').read().strip().splitlines()[-1]); c=d['config']; print('[$env] [$args]', round(d['ms_per_step'],2), c['iterations'], c['coarse_iterations'], round(c['solve_ms'],2))"
done; done
