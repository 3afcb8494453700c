#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2_exp5
mkdir -p $O; rm -f $O/log.txt
VARIANTS=r11w3,r22w3,r22w2,r33w2 python tools/exp_variants.py >> $O/log.txt 2>&1
for v in r11w3 r22w2 r33w2; do
  FI_HIP_LIB=$GRAFT_REPO_ROOT/exp_libs/libfi_$v.so python tools/exp_plain512.py >> $O/log.txt 2>&1
done
cat $O/log.txt
