#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2_exp10
mkdir -p $O; rm -f $O/*.json
for ct in 1e-5 3e-6 1e-6; do for r in 20 30; do for p in 4 5; do
  python bench.py --steps 3 --warmup 1 --cpu-side 0 --no-accuracy --levels 1 --coarse-tol $ct --poly-ratio $r --poly $p > $O/c${ct}_r${r}_p${p}.json 2>>$O/err.log
done; done; done
for f in $O/*.json; do python - "$f" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c=d['config']
print(sys.argv[1].split('/')[-1], "ms/step %.2f it %d coarse %d asm %.2f solve %.2f"%(d['ms_per_step'],c['iterations'],c['coarse_iterations'],c['assemble_ms'],c['solve_ms']))
PY
done
