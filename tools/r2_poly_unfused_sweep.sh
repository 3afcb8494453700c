#!/bin/bash
# round 2, experiment 2: Chebyshev-polynomial preconditioned CG (no coarse correction), cascade start: iteration counts
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2_exp2
mkdir -p $O
for deg in 2 3 4 6 8; do
  for ratio in 10 30 100 300; do
    FI_MG_POLY=1 FI_MG_DEGREE=$deg FI_MG_RATIO=$ratio python bench.py --steps 2 --warmup 1 --cpu-side 0 --multigrid --levels 2 > $O/poly_d${deg}_r${ratio}.json 2>> $O/err.log || echo "fail d$deg r$ratio"
  done
done
echo done
