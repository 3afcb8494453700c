#!/bin/bash
# tools/r5_sweep_levels.sh: configs 2 / 3 / 5 over hierarchy depth and level tolerance with the small-level engine (fi_tail.hip)
cd "$GRAFT_REPO_ROOT" || exit 1
run() {
  python bench.py --config $1 --levels $2 --coarse-tol $3 --steps ${4:-10} --no-cold --no-accuracy --cpu-side 0 2>&1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config', $1, 'levels', $2, 'coarse_tol', $3, 'no_tail', '${FI_NO_TAIL:-0}', 'ms', round(d['ms_per_step'],3), 'it', d['config']['iterations'], 'coarse_it', d['config']['coarse_iterations'], 'asm', round(d['config']['assemble_ms'],2), 'solve', round(d['config']['solve_ms'],2))"
}
for spec in "$@"; do run $spec; done
