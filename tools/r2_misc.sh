#!/bin/bash
# the side table of DESIGN section 5: other configurations and precision modes of the current build
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2_ev_misc; mkdir -p $O
python bench.py --steps 3 --warmup 1 --cpu-side 0 --config 2 > $O/config2.json 2>$O/err.log
python bench.py --steps 3 --warmup 1 --cpu-side 0 --config 3 > $O/config3.json 2>>$O/err.log
python bench.py --steps 3 --warmup 1 --cpu-side 0 --poly 0 --levels 2 > $O/c4_jacobi.json 2>>$O/err.log
python bench.py --steps 3 --warmup 1 --cpu-side 0 --dtype f64 > $O/c4_f64_tol1e-5.json 2>>$O/err.log
python bench.py --steps 3 --warmup 1 --cpu-side 0 --dtype f64 --tol 1e-8 > $O/c4_f64_tol1e-8.json 2>>$O/err.log
python bench.py --steps 3 --warmup 1 --cpu-side 0 --dtype f64 --tol 1e-9 > $O/c4_f64_tol1e-9.json 2>>$O/err.log
python bench.py --steps 3 --warmup 1 --cpu-side 0 --tol 1e-6 > $O/c4_f32_tol1e-6.json 2>>$O/err.log
for f in $O/*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c=d['config']
    print(sys.argv[1].split('/')[-1], "ms/step %.2f value %.3g it %d coarse %d asm %.2f solve %.2f true %.2e err %s roof %.3f apply %.3f"%(d['ms_per_step'],d['value'],c['iterations'],c['coarse_iterations'],c['assemble_ms'],c['solve_ms'],c['true_rel_residual'],d.get('solution_rel_err'),d['roofline']['frac'],d['roofline_apply']['frac']))
except Exception as e: print(sys.argv[1], 'ERR', e)
PY
done
