#!/bin/bash
# Scratch: builds exp_libs/libfi_<name>.so with extra -D flags on fi_solver.hip (experiments only; FI_HIP_LIB selects it).
set -e
cd "$(dirname "$0")/../field_interpolation_amd/csrc"
name=$1; shift
mkdir -p ../../exp_libs
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wall -Wno-unused-function "$@" \
  -c fi_solver.hip -o ../../exp_libs/fi_solver_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 fi_pool.o fi_assembly.o fi_operator.o fi_stencil.o fi_stencil2d.o fi_generic.o ../../exp_libs/fi_solver_$name.o fi_comm.o \
  -shared -Wl,-rpath,/opt/rocm/lib -ldl -lpthread -o ../../exp_libs/libfi_$name.so
echo built exp_libs/libfi_$name.so
