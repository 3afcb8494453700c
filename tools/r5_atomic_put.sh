#!/bin/bash
# scatter by LDS atomics without return (profiles/r5_ablation.md)
for side in 256 512; do
for lib in "" at64 at3264; do
  for rep in 1 2; do
  if [ -z "$lib" ]; then NAME=base SIDE=$side python tools/r4_apply_time.py || exit 1
  else NAME=$lib SIDE=$side FI_HIP_LIB=$PWD/field_interpolation_amd/libfi_$lib.so python tools/r4_apply_time.py || exit 1; fi
  done
done
done
