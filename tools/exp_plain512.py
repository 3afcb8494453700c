#!/usr/bin/env python3
"""Scratch experiment: plain (no data) AtA apply at 512^3 per dtype, tile shape (FI_TXT) and chunk length (FI_ZC)."""
import os
import subprocess
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys
sys.path.insert(0, %r)
import field_interpolation_amd as fi
side = int(os.environ["SIDE"]); dtype = os.environ["DTYPE"]
f = fi.LatticeField([side, side, side], dtype=dtype)
f.add_field_constraints(fi.Weights())
f.assemble()
f.time_apply(10)
ms = min(f.time_apply(30) for _ in range(3))
st = f.stats()
print("side %%d %%s txt %%-4s zc %%-4s: apply %%.1f us  (%%.0f GB/s algorithmic)" %% (side, dtype, os.environ.get("FI_TXT", "auto"), os.environ.get("FI_ZC", "auto"), ms * 1e3, st["spmv_bytes"] / ms / 1e6), flush=True)
''' % ROOT
for dtype in os.environ.get("DTYPES", "f64").split(","):
    for txt in os.environ.get("TXTS", "auto,16").split(","):
        for zc in os.environ.get("ZCS", "auto,64,128,256").split(","):
            env = dict(os.environ, SIDE="512", DTYPE=dtype)
            env.pop("FI_ZC", None); env.pop("FI_TXT", None)
            if zc != "auto":
                env["FI_ZC"] = zc
            if txt != "auto":
                env["FI_TXT"] = txt
            subprocess.run([sys.executable, "-c", CHILD], env=env, check=False)
