#!/bin/bash
cd "$GRAFT_REPO_ROOT"
FI_HIP_LIB=$GRAFT_REPO_ROOT/exp_libs/libfi_xpst.so timeout -k 5 120 python - <<'PY'
import ctypes as C, numpy as np, sys
sys.path.insert(0,'.')
import field_interpolation_amd as fi
from field_interpolation_amd import synth, _capi
sizes,w,pos,val=synth.config4(side=64,num_points=15625,seed=3)
f=fi.LatticeField(sizes,dtype="f32"); f.add_field_constraints(w)
f.add_points(w.data_pos,w.value_kernel,0.0,w.gradient_kernel,pos,None,None,values=val); f.assemble()
f.set_polynomial(4)
x,it,rel=f.solve_cg(None,36,1e-30)
buf=(C.c_ulonglong*8)()
assert _capi.lib().fi_debug_xp_stamps(buf)==0
print("it",it,"block0:",[buf[i] for i in range(4)],"first in->last in",buf[5]-buf[4],"first in->last out",buf[6]-buf[4])
PY
