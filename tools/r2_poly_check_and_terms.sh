#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2_exp7
mkdir -p $O; rm -f $O/*.json
python - > $O/check.txt 2>&1 <<'PY'
import numpy as np, sys
sys.path.insert(0,'.')
import field_interpolation_amd as fi
from field_interpolation_amd import synth
for side in (40, 64):
    sizes,w,pos,val=synth.config4(side=side,num_points=int(1e6*(side/256)**3),seed=3)
    for dt in ("f32","f64"):
        f=fi.LatticeField(sizes,dtype=dt); f.add_field_constraints(w)
        f.add_points(w.data_pos,w.value_kernel,0.0,w.gradient_kernel,pos,None,None,values=val); f.assemble()
        x0,it0,rel0=f.solve_cg(None,0,1e-6 if dt=="f32" else 1e-10)
        for terms in (2,3,4,6):
            f.set_polynomial(terms)
            x1,it1,rel1=f.solve_cg(None,0,1e-6 if dt=="f32" else 1e-10)
            print(side,dt,"jacobi it",it0,"poly",terms,"it",it1,"rel",rel1,"true",f.true_residual(),"maxdiff",np.abs(x1-x0).max()/np.abs(x0).max(),flush=True)
        f.set_polynomial(0)
PY
for poly in 0 2 3 4 5 6 8; do
  python bench.py --steps 3 --warmup 1 --cpu-side 0 --poly $poly > $O/poly$poly.json 2>>$O/err.log
done
python bench.py --steps 3 --warmup 1 --cpu-side 0 --poly 4 --poly-ratio 30 > $O/poly4r30.json 2>>$O/err.log
python bench.py --steps 3 --warmup 1 --cpu-side 0 --poly 6 --poly-ratio 30 > $O/poly6r30.json 2>>$O/err.log
cat $O/check.txt; tail -3 $O/err.log
