#!/bin/bash
# Scratch: builds exp_libs/libfi_<name>.so with extra -D flags on fi_stencil.hip (experiments only; FI_HIP_LIB selects it).
# usage: tools/build_variant.sh <name> [-DFOO ...]
set -e
cd "$(dirname "$0")/../field_interpolation_amd/csrc"
name=$1; shift
mkdir -p ../../exp_libs
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=fast -Wall -Wno-unused-function "$@" \
  -Rpass-analysis=kernel-resource-usage -c fi_stencil.hip -o ../../exp_libs/fi_stencil_$name.o 2> ../../exp_libs/$name.usage.txt
/opt/rocm/bin/hipcc --offload-arch=gfx950 fi_pool.o fi_assembly.o fi_operator.o ../../exp_libs/fi_stencil_$name.o fi_stencil_lists.o fi_stencil2d.o fi_generic.o fi_tail.o fi_cg.o fi_poly.o fi_transfer.o fi_multigrid.o fi_levels.o fi_capi.o fi_group.o fi_comm.o \
  -shared -Wl,-rpath,/opt/rocm/lib -ldl -lpthread -o ../../exp_libs/libfi_$name.so
echo built exp_libs/libfi_$name.so
