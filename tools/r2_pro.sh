#!/bin/bash
# timing variants of the first Chebyshev step (PRO): which of its extra loads costs what (results wrong by construction)
cd "$GRAFT_REPO_ROOT"
O=$GRAFT_REPO_ROOT/gpurun_out/r2_pro; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
export FI_HIP_LIB=$GRAFT_REPO_ROOT/exp_libs/libfi_tbs.so FI_SOLVE_TIMEOUT_S=20
for dbg in 0 64 128 192 1 193; do
export FI_DBG=$dbg
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --cpu-side 0 --no-accuracy --max-iterations 24 > $O/trace.log 2>&1
echo "== FI_DBG=$dbg"; python3 $GRAFT_REPO_ROOT/tools/trace_poly_steps.py $O/trace | grep -E "step|apply"
rm -rf $O/trace
done
