#!/bin/bash
# Scratch: builds exp_libs/libfi_<name>.so with extra -D flags on EVERY source (experiments only; FI_HIP_LIB selects it).
set -e
cd "$(dirname "$0")/../field_interpolation_amd/csrc"
name=$1; shift
mkdir -p ../../exp_libs/$name
for f in fi_pool fi_assembly fi_operator fi_stencil_lists fi_stencil2d fi_generic fi_tail fi_cg fi_poly fi_transfer fi_multigrid fi_levels fi_capi fi_group fi_comm; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wall -Wno-unused-function "$@" -c $f.hip -o ../../exp_libs/$name/$f.o &
done
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=fast -Wall -Wno-unused-function "$@" -c fi_stencil.hip -o ../../exp_libs/$name/fi_stencil.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 ../../exp_libs/$name/*.o -shared -Wl,-rpath,/opt/rocm/lib -ldl -lpthread -o ../../exp_libs/libfi_$name.so
echo built exp_libs/libfi_$name.so
