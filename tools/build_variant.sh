#!/bin/bash
# Scratch: builds exp_libs/libfi_<name>.so with extra -D flags on ONE source of the library (SRC=fi_stencil by default, e.g.
# SRC=fi_strip); experiments only, FI_HIP_LIB selects the result.
# usage: [SRC=fi_strip] tools/build_variant.sh <name> [-DFOO ...]
set -e
cd "$(dirname "$0")/../field_interpolation_amd/csrc"
name=$1; shift
src=${SRC:-fi_stencil}
mkdir -p ../../exp_libs
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=fast -Wall -Wno-unused-function "$@" \
  -Rpass-analysis=kernel-resource-usage -c $src.hip -o ../../exp_libs/${src}_$name.o 2> ../../exp_libs/$name.usage.txt
objs=""
for o in fi_pool fi_assembly fi_operator fi_stencil fi_strip fi_stencil_lists fi_stencil2d fi_generic fi_tail fi_cg fi_poly fi_transfer fi_multigrid fi_levels fi_capi fi_group fi_comm; do
  if [ $o = $src ]; then objs="$objs ../../exp_libs/${src}_$name.o"; else objs="$objs $o.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 $objs -shared -Wl,-rpath,/opt/rocm/lib -ldl -lpthread -o ../../exp_libs/libfi_$name.so
echo built exp_libs/libfi_$name.so
