#!/bin/bash
# isolated fp64 apply at 512^3 / 256^3 for builds of fi_strip.hip (exp_libs/libfi_<name>.so); profiles/r6_ablation.md
for name in "$@"; do
  if [ $name = shipped ]; then unset FI_HIP_LIB; else export FI_HIP_LIB=$PWD/exp_libs/libfi_$name.so; fi
  NAME=$name DTYPES=f64 DATA=${DATA:-none,c4,c5} SIDE=${SIDE:-512} python tools/r6_apply_probe.py || exit 1
done
