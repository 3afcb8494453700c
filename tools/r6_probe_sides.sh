#!/bin/bash
# is the plain apply's 0.57 a property of the power-of-two row pitch?  (profiles/r6_ablation.md)
for side in 384 448 480 496 512 520 544 576 640; do
  SIDE=$side DATA=none DTYPES=f64,f32 python tools/r6_apply_probe.py || exit 1
done
