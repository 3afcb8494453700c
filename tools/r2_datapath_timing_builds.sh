#!/bin/bash
# timing builds of the fused apply (results wrong by construction): tools/build_variant.sh tdense -DFI_TIMING_BUILD -DFI_DENSE_MIN=0u ; ... tsparse -DFI_TIMING_BUILD -DFI_DENSE_MIN=100000u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2_exp4
mkdir -p $O; rm -f $O/log.txt
for v in tdense tsparse; do
for dbg in 0 64 72 88; do
  echo "== $v FI_DBG=$dbg" >> $O/log.txt
  FI_DBG=$dbg NOSOLVE=1 VARIANTS=$v python tools/exp_variants.py >> $O/log.txt 2>&1
done
done
cat $O/log.txt
