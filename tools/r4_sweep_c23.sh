#!/bin/bash
# tools/r4_sweep_c23.sh: configs 2 and 3 over hierarchy depth and the levels' tolerance -> gpurun_out/sweep_c23.txt
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/sweep_c23.txt; : > $out
for cfg in 3 2; do
  for lv in 5 6 7 8 9; do
    for ct in 1e-4 1e-3 1e-2; do
      python3 bench.py --config $cfg --steps 3 --warmup 1 --cpu-side 0 --no-accuracy --no-cold --levels $lv --coarse-tol $ct 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline())
c=d['config']
print('cfg $cfg levels $lv ctol $ct: %.2f ms/step  it %d coarse %d  asm %.2f solve %.2f  rel %.2e' % (d['ms_per_step'], c['iterations'], c.get('coarse_iterations',0), c['assemble_ms'], c['solve_ms'], c.get('true_rel_residual', c.get('rel_residual', 0))))" >> $out
      tail -1 $out
    done
  done
done
