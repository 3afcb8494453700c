#!/bin/bash
# terms x ratio of the polynomial preconditioner under the final per-step costs
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2_sweep2; mkdir -p $O
for t in 3 4 5 6; do for r in 15 30 50; do
  timeout -k 10 120 python bench.py --steps 3 --warmup 1 --cpu-side 0 --no-accuracy --poly $t --poly-ratio $r > $O/b.json 2> $O/b.err && python -c "
import json; d=json.loads(open('$O/b.json').read().strip().splitlines()[-1]); c=d['config']; print('terms $t ratio $r', round(d['ms_per_step'],2), c['iterations'], c['coarse_iterations'], round(c['solve_ms'],2))"
done; done
