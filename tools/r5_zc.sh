#!/bin/bash
# chunk length of the fused fp64 apply (FI_ZC) -- profiles/r5_ablation.md
for side in 256 512; do
  for zc in 0 64 52 43 37 32 26 22 16; do
    if [ $zc = 0 ]; then NAME=default SIDE=$side DTYPES=f64 python tools/r4_apply_time.py || exit 1
    else NAME=zc$zc FI_ZC=$zc SIDE=$side DTYPES=f64 python tools/r4_apply_time.py || exit 1; fi
  done
done
