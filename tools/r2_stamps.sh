#!/bin/bash
# in-kernel stamps of one workgroup of the apply: build first with  tools/build_variant.sh stamps -DFI_STAMPS -DFI_TIMING_BUILD
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2_exp6
mkdir -p $O
FI_HIP_LIB=$GRAFT_REPO_ROOT/exp_libs/libfi_stamps.so python tools/exp_stamps.py > $O/stamps_data.txt 2>&1
FI_HIP_LIB=$GRAFT_REPO_ROOT/exp_libs/libfi_stamps.so NODATA=1 python tools/exp_stamps.py > $O/stamps_plain.txt 2>&1
cat $O/stamps_data.txt $O/stamps_plain.txt
