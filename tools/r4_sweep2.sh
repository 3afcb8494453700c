#!/bin/bash
# tools/r4_sweep2.sh "<cfg> <levels> <ctol>" ...: one bench.py run per argument -> gpurun_out/sweep2.txt
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/sweep2.txt; : > $out
for spec in "$@"; do
  set -- $spec
  python3 bench.py --config $1 --steps 3 --warmup 1 --cpu-side 0 --no-accuracy --no-cold --levels $2 --coarse-tol $3 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline())
c=d['config']
print('cfg $1 levels $2 ctol $3: %.2f ms/step  it %d coarse %d  asm %.2f solve %.2f  rel %.2e' % (d['ms_per_step'], c['iterations'], c.get('coarse_iterations',0), c['assemble_ms'], c['solve_ms'], c.get('true_rel_residual', c.get('rel_residual', 0))))" >> $out
  tail -1 $out
done
