#!/bin/bash
# The full-operator Chebyshev smoother (degree, interval ratio: the levels that do not run the polynomial smoother -- oriented
# points, 2-D lattices) under the field stop rule.  Needs a timing build of fi_multigrid.hip:
#   SRC=fi_multigrid tools/build_variant.sh mgsw -DFI_TIMING_BUILD
# usage (GPU box): CONFIG=5 DEGREES="4 5 6" RATIOS="10 20 40" bash tools/r6_sweep_sdf_smoother.sh [bench args]
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r6
export FI_HIP_LIB=$PWD/exp_libs/libfi_mgsw.so
cfg=${CONFIG:-5}
for deg in ${DEGREES:-3 4 5 6}; do
  for ratio in ${RATIOS:-10 20 40}; do
    out=gpurun_out/r6/smoother_c${cfg}_d${deg}_r${ratio}.json
    FI_MG_DEGREE=$deg FI_MG_RATIO=$ratio python bench.py --config $cfg --steps 2 --warmup 1 --no-cold --cpu-side 0 "$@" > $out 2>/dev/null || { echo "config $cfg degree $deg ratio $ratio failed"; continue; }
    echo -n "config $cfg degree $deg ratio $ratio: "; python3 tools/bench_brief.py $out
  done
done
