#!/bin/bash
# polynomial of the cascade's coarse level (config 4, 256^3): terms / ratio there, the finest level as shipped
cd "$GRAFT_REPO_ROOT"
export FI_HIP_LIB=$PWD/exp_libs/libfi_timing.so NOREF=1
for tr in "4 30" "6 60" "8 100" "8 60" "10 150" "12 200" "6 30"; do
  set -- $tr
  echo "coarse terms $1 ratio $2" 
  FI_COARSE_TERMS=$1 FI_COARSE_RATIO=$2 MODES="a:f32:1:0:0:4:30:1e-5" CTOL=1e-5 timeout -k 10 120 python tools/exp.py 2>&1 | grep "ms/step"
done
